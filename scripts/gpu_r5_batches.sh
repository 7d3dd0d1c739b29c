#!/bin/bash
# round 5: bench.py at several minibatch sizes on ONE box (throughput-regime work).
# usage: gpurun --timeout 900 -- 'BATCHES="1024 4096" ENVS="A=0|DSACT_X=1" bash scripts/gpu_r5_batches.sh ["<pytest -k expr>"]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r5_batches; rm -rf $OUT; mkdir -p $OUT
timeout 400 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
summ() { grep "^{\"metric\"" $1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.2f  finite %s  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d.get('finite_stats'), ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))
except Exception as e:
    print('   parse error', e)
"; }
IFS='|' read -ra EV <<< "${ENVS:-A=0}"
for b in ${BATCHES:-1024}; do
  i=0
  for e in "${EV[@]}"; do
    i=$((i+1))
    env $e timeout 300 python bench.py --batch $b --steps ${BENCH_STEPS:-1000} --warmup ${BENCH_WARMUP:-200} --no-cpu-baseline --no-alt > $OUT/bench_b${b}_$i.log 2>&1
    echo "== batch $b ($e) rc=$?"; summ $OUT/bench_b${b}_$i.log
  done
done
K="${1:-}"
if [ -n "$K" ]; then
  timeout 1400 python -m pytest ${TESTS:-tests} -q -m gpu -p no:cacheprovider --timeout 600 -x -k "$K" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  grep -n "^FAILED\|^ERROR\|passed\|failed\|^E  " $OUT/pytest_gpu.log | tail -30
fi
