#!/bin/bash
# round 3: targeted GPU tests + A/B benches of env switches. usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r3_ab.sh "<pytest -k expr>" "<ENV=1 ...>" ["<ENV2=1>" ...]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_ab; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
K="${1:-}"; shift
if [ -n "$K" ]; then
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "$K" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_gpu.log
fi
STEPS=${BENCH_STEPS:-4000}
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps $STEPS --warmup 400 --no-cpu-baseline --no-alt ${BENCH_ARGS:-} > $OUT/bench_$label.log 2>&1
  echo "== $label ($*) rc=$?"
  tail -1 $OUT/bench_$label.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  ms %.4f  kernels %s' % (d['value'], d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
except Exception as e:
    print('   parse error', e)
"
}
run default A=0
i=0
for e in "$@"; do i=$((i+1)); run alt$i $e; done
run default2 A=0
