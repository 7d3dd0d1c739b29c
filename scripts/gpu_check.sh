#!/bin/bash
# One gpurun call: GPU parity tests (full report), smoke, short bench, rocprofv3 kernel stats.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh [quick]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/parity_report.txt
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
echo "== build" | tee $OUT/steps.log
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1; echo "build rc=$?" | tee -a $OUT/steps.log
echo "== pytest gpu" | tee -a $OUT/steps.log
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/steps.log
tail -40 $OUT/pytest_gpu.log
echo "== smoke" | tee -a $OUT/steps.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $OUT/steps.log
tail -3 $OUT/smoke.log
if [ "${1:-}" != "quick" ]; then
echo "== bench" | tee -a $OUT/steps.log
timeout 600 python bench.py --steps 4000 --warmup 400 > $OUT/bench.log 2>&1; echo "bench rc=$?" | tee -a $OUT/steps.log
tail -2 $OUT/bench.log
echo "== rocprof" | tee -a $OUT/steps.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?" | tee -a $OUT/steps.log
ls -R $OUT/prof | head -20
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] && python scripts/step_trace.py "$f" 1000 > $OUT/step_trace.txt
fi
