#!/bin/bash
# k-tiles per k_conv_dw workgroup, per layer (DSACT_CONV_DW_NKT_L): per-launch times of the conv weight gradients
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in ${NKTS:-"1,1,1,1,1,1" "2,1,2,2,2,2" "2,2,2,2,2,2" "2,1,3,2,2,2" "2,1,2,3,3,3"}; do
  DSACT_CONV_DW_NKT_L=$v timeout 300 python bench.py --cnn-only --cnn-steps 300 --no-cpu-baseline 2>/dev/null | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['cnn']
print('$v', round(d['value'], 1), 'steps/s', ' '.join('%s=%.1f' % (k['name'][5:], k['us']) for k in d['kernels'] if k['name'].startswith('conv_dw')))"
done 2>&1 | tee gpurun_out/r04_convdw_nkt.txt
