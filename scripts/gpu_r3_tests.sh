#!/bin/bash
# round 3: GPU tests (+ parity report) + the driver's bench command. usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r3_tests.sh [pytest -k expr]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_tests; rm -rf $OUT; mkdir -p $OUT; rm -f gpurun_out/parity_report.txt
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ -n "${1:-}" ]; then
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -x -k "$1" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
else
  timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
fi
tail -30 $OUT/pytest_gpu.log
cp gpurun_out/parity_report.txt $OUT/parity_report.txt 2>/dev/null
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver.log 2>&1; echo "bench rc=$?"; tail -1 $OUT/bench_driver.log | cut -c1-400
