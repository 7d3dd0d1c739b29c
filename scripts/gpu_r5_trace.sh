#!/bin/bash
# rocprofv3 kernel trace of the strict bench only (--fast: skip) -> launch-by-launch view of two consecutive updates
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_trace; rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python scripts/step_trace.py "$f" 1001 > $OUT/step_trace.txt; cat $OUT/step_trace.txt
python scripts/step_trace.py "$f" 1003 | tail -20 > $OUT/step_trace_b.txt
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
rm -rf $OUT/prof
