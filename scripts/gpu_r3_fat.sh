#!/bin/bash
# round 3: throughput-regime kernels -- parity tests, then bench per batch size with / without them.
# usage: gpurun --timeout 1500 -- 'bash scripts/gpu_r3_fat.sh ["<pytest -k expr>"] ["<batch sizes>"]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r3_fat; rm -rf $OUT; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
K="${1:-throughput or b512 or b2048 or split_k or large_batch}"
if [ "$K" != "none" ]; then
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "$K" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 $OUT/pytest_gpu.log
fi
run() { # label, batch, env...
  local label=$1 b=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 1000 --warmup 100 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_$label.log 2>&1
  echo "== $label ($*) rc=$?"
  tail -1 $OUT/bench_$label.log | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('   value %.0f  us %.1f  step_frac %.3f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], d.get('roofline_step', {}).get('frac', -1), ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', []))))
except Exception as e:
    print('   parse error', e)
"
}
for b in ${2:-512 1024 2048 4096}; do
  run b${b}_default $b A=0
  run b${b}_nofat $b DSACT_NO_FAT=1
  run b${b}_fat512_rt1 $b DSACT_FAT_MIN=512 DSACT_FAT_RT=1
  run b${b}_fat512_rt2 $b DSACT_FAT_MIN=512 DSACT_FAT_RT=2
done
