#!/bin/bash
# register-tile weight gradient of the narrow conv layers (DSACT_CONV_DW_REG=1): CNN parity tests with it, then the CNN leg A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out; export TMPDIR=/tmp
if [ "$1" = test ]; then
  DSACT_CONV_DW_REG=${REGMASK:-7} timeout 900 python -m pytest tests/test_hip_cnn_parity.py tests/test_hip_v1_cnn_parity.py -m gpu -q 2>&1 | tail -12
fi
run() { env "$@" timeout 300 python bench.py --cnn-only --cnn-steps 300 --no-cpu-baseline 2>/dev/null | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.readline())['cnn']
print('$*', round(d['value'], 1), 'steps/s', ' '.join('%s=%.1f' % (k['name'][5:], k['us']) for k in d['kernels'] if k['name'].startswith('conv_dw')))"; }
for v in ${VARIANTS:-"A=0" "DSACT_CONV_DW_REG=2" "DSACT_CONV_DW_REG=2 DSACT_CONV_DW_REG_WGS=1024" "DSACT_CONV_DW_REG=2 DSACT_CONV_DW_REG_WGS=2048" "DSACT_CONV_DW_REG=3 DSACT_CONV_DW_REG_WGS=1024" "DSACT_CONV_DW_REG=7 DSACT_CONV_DW_REG_WGS=1024"}; do run $v; done
