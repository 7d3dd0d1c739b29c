#!/bin/bash
# kernel trace of the driver-style short run (--steps 20 --warmup 5): where do a short region's microseconds go?
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT; rm -rf $OUT/prof_short
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_short -o bench -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt "$@" > $OUT/rocprof_short.log 2>&1; echo "rocprof rc=$?"
tail -1 $OUT/rocprof_short.log | cut -c1-300
f=$(find $OUT/prof_short -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 40 --bursts > $OUT/step_trace_short.txt
cat $OUT/step_trace_short.txt
rm -rf $OUT/prof_short
