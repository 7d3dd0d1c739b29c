#!/bin/bash
# quick GPU iteration: build, a few parity tests, a short bench.  usage: gpurun -- 'bash scripts/gpu_quick.sh "<pytest -k expr>" [bench args]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/parity_report.txt
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
K="${1:-humanoid_l3}"; shift || true
timeout 900 python -m pytest tests -q -x -m gpu -p no:cacheprovider --timeout 600 -k "$K" > $OUT/pytest_quick.log 2>&1; echo "pytest rc=$?"
tail -45 $OUT/pytest_quick.log
if [ $# -gt 0 ]; then
  timeout 600 python bench.py "$@" > $OUT/bench_quick.log 2>&1; echo "bench rc=$?"; tail -3 $OUT/bench_quick.log | cut -c1-1500
fi
