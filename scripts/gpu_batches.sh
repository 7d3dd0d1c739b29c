#!/bin/bash
# bench at several batch sizes (+ optional env A/B as $1): usage: gpurun -- 'bash scripts/gpu_batches.sh [NAME=VALUE]'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python __graft_entry__.py > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for b in 256 512 1024 4096; do
  for e in "X_=1" "$@"; do
    env $e timeout 300 python bench.py --steps 1000 --warmup 100 --batch $b --no-cpu-baseline --no-alt 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('batch $b [$e]: %.0f steps/s  %.1f us  step frac %.3f  kernels %s' % (d['value'], d['ms_per_step'] * 1e3, d['roofline_step']['frac'], ' '.join('%s=%.1f' % (k['name'].replace('chain_', ''), k['us']) for k in d.get('kernels', []))))"
  done
done
