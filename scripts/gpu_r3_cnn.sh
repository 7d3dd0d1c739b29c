#!/bin/bash
# round 3: CNN workload A/B. usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r3_cnn.sh "<ENV=1>" [...]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_cnn; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
run() { local label=$1; shift
  env "$@" timeout 300 python bench.py --cnn-only --cnn-steps 300 --no-cpu-baseline > $OUT/bench_$label.log 2>&1
  echo "== $label ($*) rc=$?"
  grep "^{" $OUT/bench_$label.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
d = d.get('cnn', d)
print('   steps/s %.0f  us %.1f  launches %d' % (d['value'], 1000 * d['ms_per_step'], len(d.get('kernels', []))))
print('   ' + ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', [])))
"; }
run default A=0
i=0
for e in "$@"; do i=$((i+1)); run alt$i $e
  env $e timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "cnn" > $OUT/pytest_alt$i.log 2>&1; echo "pytest ($e) rc=$?"; tail -2 $OUT/pytest_alt$i.log
done
run default2 A=0
