#!/bin/bash
# round 5: targeted GPU tests + A/B benches of env switches on ONE box.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r5_b.sh "<pytest -k expr>" "<ENV=1 ...>" ["<ENV2=1>" ...]'
#   BENCH_STEPS (default 4000), BENCH_WARMUP (400), BENCH_ARGS, DRIVER=1 (also the driver's --steps 20 --warmup 5 command per variant), TESTS="files"
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r5_b; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
K="${1:-}"; shift
if [ -n "$K" ]; then
  timeout 1400 python -m pytest ${TESTS:-tests} -q -m gpu -p no:cacheprovider --timeout 600 -x -k "$K" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_gpu.log | tail -30
fi
STEPS=${BENCH_STEPS:-4000}; WARM=${BENCH_WARMUP:-400}
summ() { grep "^{\"metric\"" $1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  finite %s  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d.get('finite_stats'), ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
except Exception as e:
    print('   parse error', e)
"; }
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-alt ${BENCH_ARGS:-} > $OUT/bench_$label.log 2>&1
  echo "== $label ($*) rc=$?"; summ $OUT/bench_$label.log
  if [ -n "${DRIVER:-}" ]; then
    env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt ${BENCH_ARGS:-} > $OUT/bench_${label}_drv.log 2>&1
    echo "   driver command rc=$?"; summ $OUT/bench_${label}_drv.log
  fi
}
run default A=0
i=0
for e in "$@"; do i=$((i+1)); run alt$i $e; done
run default2 A=0
