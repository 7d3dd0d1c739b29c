#!/bin/bash
# thresholds of the 64 x 64 conv forward / dCol tiles: layer 5 (96 forward tiles, 256-row dCol) on them or not
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "256 512" "64 512" "256 256" "64 256"; do
  set -- $v
  DSACT_CONV_FWD64_MIN=$1 DSACT_DCOL64_MIN_M=$2 timeout 300 python bench.py --cnn-only --cnn-steps 300 --no-cpu-baseline 2>/dev/null | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['cnn']
print('fwd64_min=$1 dcol64_min_m=$2', round(d['value'], 1), 'steps/s', ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d['kernels'] if k['name'] in ('conv_fwd_l5', 'conv_fwd_l4', 'conv_dcol_l5', 'conv_dcol_l4')))"
done 2>&1 | tee gpurun_out/r04_tile64_min.txt
