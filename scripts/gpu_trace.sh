#!/bin/bash
# rocprofv3 kernel trace of a short bench run -> per-launch step trace. usage: gpurun -- 'bash scripts/gpu_trace.sh [bench args]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT; rm -rf $OUT/prof
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt "$@" > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 1000 > $OUT/step_trace.txt
cat $OUT/step_trace.txt | head -24
head -8 $OUT/prof/bench_kernel_stats.csv | cut -c1-160
rm -f $OUT/prof/bench_kernel_trace.csv
