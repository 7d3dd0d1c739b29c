#!/bin/bash
# round 6: output activations on the row-slice chains -- their tests, the launch-form tests, then the headline A/B against HEAD~ is not
# possible on one tree: the default kernels' register counts are unchanged (scripts/kernel_meta.py), the bench confirms.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_f; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 \
  -k "output_activation or host_acting or pipelined or std_param or parameter_std or unequal or act_sample or humanoid or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest.log | tail -25
for i in 1 2; do
timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_long_$i.log 2>&1
grep '^{"metric"' $OUT/bench_long_$i.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   long run %d: value %.0f  us %.2f' % ($i, d['value'], 1000 * d['ms_per_step']))"
done
