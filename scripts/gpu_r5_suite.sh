#!/bin/bash
# the whole -m gpu suite + smoke (what the driver runs at round end); usage: gpurun --timeout 2400 -- 'bash scripts/gpu_r5_suite.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r5_suite; rm -rf $OUT; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 2000 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_gpu.log | tail -40
cp gpurun_out/parity_report.txt $OUT/parity_report.txt 2>/dev/null
timeout 600 python -c "import __graft_entry__ as e; e.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/smoke.log
