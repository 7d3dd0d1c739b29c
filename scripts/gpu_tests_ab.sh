#!/bin/bash
# full GPU test suite + bench; then an A/B rebuild with extra hipcc flags ($1) and the same bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/parity_report.txt
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "${SKIP_TESTS:-0}" != 1 ]; then
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 $OUT/pytest_gpu.log | cut -c1-400
fi
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $OUT/bench_driver.log 2>&1; echo "driver-style rc=$?"
tail -1 $OUT/bench_driver.log | cut -c1-420; grep -o '"kernels": \[[^]]*\]' $OUT/bench_driver.log | head -1
timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/bench_long.log 2>&1; echo "long rc=$?"; tail -1 $OUT/bench_long.log | cut -c1-420
if [ $# -gt 0 ]; then
  DSACT_HIPCC_EXTRA="$1" timeout 300 python -c "import __graft_entry__ as g; g.build(force=True)" > $OUT/build_b.log 2>&1 || { tail -20 $OUT/build_b.log; exit 1; }
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $OUT/bench_driver_b.log 2>&1; echo "B driver-style rc=$?"
  tail -1 $OUT/bench_driver_b.log | cut -c1-420; grep -o '"kernels": \[[^]]*\]' $OUT/bench_driver_b.log | head -1
  timeout 300 python bench.py --steps 4000 --warmup 200 --no-cpu-baseline --no-alt > $OUT/bench_long_b.log 2>&1; echo "B long rc=$?"; tail -1 $OUT/bench_long_b.log | cut -c1-420
fi
