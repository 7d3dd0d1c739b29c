#!/bin/bash
# data-parallel legs at world 1: tests + forced-collective bench (native graph, eager)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "native_rccl or data_parallel or merged_forward or strict_data" > $OUT/pytest_dp.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_dp.log | cut -c1-250
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/bench_dp_native2.log 2>&1; echo "dp native rc=$?"; tail -4 $OUT/bench_dp_native2.log | cut -c1-400
