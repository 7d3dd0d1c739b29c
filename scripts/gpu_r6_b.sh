#!/bin/bash
# round 6: the q(obs, new_act) backward riding in the policy-moving update's forward launch (kPipeRoleBwdQp) -- bitwise tests of the
# launch forms, then A/B against DSACT_NO_QPB=1 (round 5's form) on one box. usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r6_b.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_b; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 900 -x \
  -k "${TESTS:-pipelined or poison or handover or run_group or fast_mode or graph_replay or host_acting or cnn_si8}" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest.log | tail -20
summ() { grep "^{\"metric\"" $1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  finite %s  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d.get('finite_stats'), ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
except Exception as e:
    print('   parse error', e)
"; }
run() { # label, env assignment or ""
  env $2 timeout 300 python bench.py --steps ${BENCH_STEPS:-4000} --warmup ${BENCH_WARMUP:-400} --no-cpu-baseline --no-alt ${BENCH_ARGS:-} > $OUT/bench_$1.log 2>&1
  echo "== $1 rc=$?"; summ $OUT/bench_$1.log
}
run qpb "X=1"
run noqpb "DSACT_NO_QPB=1"
run qpb2 "X=1"
run actor_only "DSACT_QPB_ACTOR_ONLY=1"
BENCH_STEPS=20 BENCH_WARMUP=5 run qpb_driver "X=1"
BENCH_STEPS=20 BENCH_WARMUP=5 run noqpb_driver "DSACT_NO_QPB=1"
