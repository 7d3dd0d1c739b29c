#!/bin/bash
# validates bench.py the way the driver calls it, plus the forced data-parallel legs at world 1
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
t0=$(date +%s.%N)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2>&1; echo "driver-style rc=$? wall=$(echo "$(date +%s.%N) - $t0" | bc)"
tail -1 $OUT/bench_driver.log | cut -c1-2500
timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu --no-alt > $OUT/bench_long.log 2>&1; echo "long rc=$?"; tail -1 $OUT/bench_long.log | cut -c1-600
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu --no-alt > $OUT/bench_dp_native.log 2>&1; echo "dp native rc=$?"; tail -1 $OUT/bench_dp_native.log | cut -c1-600
DSACT_BENCH_FORCE_DP=1 timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu --no-alt --dp-eager > $OUT/bench_dp_eager.log 2>&1; echo "dp eager rc=$?"; tail -1 $OUT/bench_dp_eager.log | cut -c1-600
