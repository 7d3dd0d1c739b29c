#!/bin/bash
# CNN workload (BASELINE.json configs[3]): bench object + rocprofv3 kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 300 python bench.py --cnn-only --cnn-steps 200 > $OUT/bench_cnn.log 2>&1; echo "bench_cnn rc=$?"
tail -c 6000 $OUT/bench_cnn.log
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cnn -o cnn -- python bench.py --cnn-only --cnn-steps 100 --no-cpu-baseline > $OUT/rocprof_cnn.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_cnn -name "*kernel_stats*" | head
