#!/bin/bash
# full GPU suite + smoke + the driver's bench command on the current tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/suite; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > gpurun_out/suite/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/suite/pytest_gpu.log
cp gpurun_out/parity_report.txt gpurun_out/suite/parity_report.txt 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > gpurun_out/suite/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/suite/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/suite/bench_driver.log 2>&1; echo "driver bench rc=$?"; grep '^{"metric"' gpurun_out/suite/bench_driver.log | cut -c1-220
