#!/bin/bash
# rocprofv3 kernel-trace stats of a short bench run (+ optional PMC pass). usage: gpurun -- 'bash scripts/gpu_prof.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
python __graft_entry__.py > $OUT/build.log 2>&1 || { tail $OUT/build.log; exit 1; }
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/rocprof.log 2>&1
echo "rocprof rc=$?"
tail -1 $OUT/rocprof.log | cut -c1-400
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && column -s, -t "$f" | cut -c1-200 | head -30
