#!/bin/bash
# round 6: placement / rows-per-workgroup variants of the riding critic backward (zero-code: DSACT_PIPE_MAP), long runs on one box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_c; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
summ() { grep "^{\"metric\"" $1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  finite %s  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d.get('finite_stats'), ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []) if k['name'] in ('chain_fwd+next','chain_bwd_qt','chain_fwd_q','chain_bwd_qpt'))))
except Exception as e:
    print('   parse error', e)
"; }
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt > $OUT/bench_$label.log 2>&1
  echo "== $label rc=$? ($*)"; summ $OUT/bench_$label.log
}
M4="TF.q1p=0246:1;TF.q2p=1357:1;TF.q1c=04:2;TF.q2c=15:2;TF.q1t=26:2;TF.q2t=37:2"
run base DSACT_NO_QPB=1
run base_map4 DSACT_NO_QPB=1 "DSACT_PIPE_MAP=$M4"
run actor DSACT_QPB_ACTOR_ONLY=1
run actor_map4 DSACT_QPB_ACTOR_ONLY=1 "DSACT_PIPE_MAP=$M4"
run all4_map4 X=1 "DSACT_PIPE_MAP=$M4"
M5="TF.q1p=0246:1;TF.q2p=1357:1;TF.q1c=0246:2;TF.q2c=1357:2;TF.q1t=0246:2;TF.q2t=1357:2"
run actor_map5 DSACT_QPB_ACTOR_ONLY=1 "DSACT_PIPE_MAP=$M5"
run base2 DSACT_NO_QPB=1
