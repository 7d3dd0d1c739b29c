#!/bin/bash
# launch-by-launch trace of one CNN update (rocprofv3 --kernel-trace only). usage: gpurun --timeout 900 -- 'bash scripts/gpu_r3_cnn_trace.sh [ENV=1 ...]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_cnn_trace; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
env "$@" timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o cnn -- python bench.py --cnn-only --cnn-steps 100 --no-cpu-baseline > $OUT/rocprof_cnn.log 2>&1; echo "rocprof cnn rc=$?"
tail -1 $OUT/rocprof_cnn.log | cut -c1-300
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 50 --close=k_conv_dw_reduce > $OUT/cnn_step_trace.txt && head -60 $OUT/cnn_step_trace.txt
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/cnn_kernel_stats.csv 2>/dev/null
rm -rf $OUT/prof
