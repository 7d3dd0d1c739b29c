#!/bin/bash
# rocprofv3 kernel stats of the forced data-parallel graph at world 1: what the step consists of
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/dp_trace; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || exit 1
DSACT_BENCH_FORCE_DP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o dp -- python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-alt > $OUT/run.log 2>&1; echo "rc=$?"
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-160 | tee $OUT/dp_kernel_stats_head.txt
cp "$f" $OUT/dp_kernel_stats.csv; rm -rf $OUT/p
