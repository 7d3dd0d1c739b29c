#!/bin/bash
# round 6: CNN -- layer 0 reading the ring rows in place (ImgIndex) vs the staged copy (DSACT_NO_IMG_DIRECT=1): every CNN test in
# both modes, then the bench object of each. usage: gpurun --timeout 1800 -- 'bash scripts/gpu_r6_cnn.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_cnn; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
K="cnn and not end_to_end"
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "$K" > $OUT/pytest_direct.log 2>&1; echo "pytest (direct) rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_direct.log | tail -15
if [ "${SKIP_STAGED:-0}" != "1" ]; then
DSACT_NO_IMG_DIRECT=1 timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "$K" > $OUT/pytest_staged.log 2>&1; echo "pytest (staged) rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed" $OUT/pytest_staged.log | tail -5
fi
for m in direct staged direct2 staged2; do
  if [ "${m#staged}" != "$m" ]; then export DSACT_NO_IMG_DIRECT=1; else unset DSACT_NO_IMG_DIRECT; fi
  timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline > $OUT/bench_cnn_$m.log 2>&1; echo "cnn bench $m rc=$?"
  grep '^{"cnn"' $OUT/bench_cnn_$m.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())['cnn']
print('   %.0f steps/s  %.1f us   %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', [])[:8])))"
done
unset DSACT_NO_IMG_DIRECT
