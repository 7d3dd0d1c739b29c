#!/bin/bash
# k-tiles per k_conv_dw workgroup, per layer (DSACT_CONV_DW_NKT_L), and the single-buffered three-k-tile form (DSACT_CONV_DW_SB3)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --cnn-only --cnn-steps 300 --no-cpu-baseline 2>/dev/null | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['cnn']
print('$*', round(d['value'], 1), 'steps/s', ' '.join('%s=%.1f' % (k['name'][5:], k['us']) for k in d['kernels'] if k['name'].startswith('conv_dw')))"; }
run A=0
run DSACT_CONV_DW_SB3=1 DSACT_CONV_DW_NKT_L=2,3,2,2,2,2
run DSACT_CONV_DW_SB3=1 DSACT_CONV_DW_NKT_L=2,3,3,2,2,2
run DSACT_CONV_DW_SB3=1 DSACT_CONV_DW_NKT_L=2,3,3,3,3,3
