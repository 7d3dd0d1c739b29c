#!/bin/bash
# round 5: A/B of differently COMPILED libraries (build/libdsact_*.so, built in the container) on ONE box.
# usage: gpurun --timeout 900 -- 'LIBS="nap4 nap10" bash scripts/gpu_r5_libs.sh'   (BENCH_STEPS / BENCH_WARMUP / BENCH_ARGS as in gpu_r5_b.sh)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r5_libs; rm -rf $OUT; mkdir -p $OUT
STEPS=${BENCH_STEPS:-4000}; WARM=${BENCH_WARMUP:-400}
summ() { grep "^{\"metric\"" $1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  finite %s  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d.get('finite_stats'), ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
except Exception as e:
    print('   parse error', e)
"; }
run() { # label, lib path or ""
  if [ -n "$2" ]; then export DSACT_LIB_PATH=$2; else unset DSACT_LIB_PATH; fi
  timeout 300 python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-alt ${BENCH_ARGS:-} > $OUT/bench_$1.log 2>&1
  echo "== $1 rc=$?"; summ $OUT/bench_$1.log
}
run default ""
for l in ${LIBS:-}; do run $l $PWD/build/libdsact_$l.so; done
run default2 ""
for l in ${LIBS:-}; do run ${l}_2 $PWD/build/libdsact_$l.so; done
