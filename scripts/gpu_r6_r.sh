#!/bin/bash
# round 6: throughput-regime forward, a finished tile's epilogue under the MFMAs of the tiles behind it (last two chunks tile by tile) -- tests + A/B vs build/libdsact_prev.so
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -p no:cacheprovider -k "throughput_regime" 2>&1 | tail -3
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps $4 --warmup 200 --batch $3 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
P="DSACT_LIB_PATH=$PWD/build/libdsact_prev.so"
{
run new_1024 "X=1" 1024 1000; run prev_1024 "$P" 1024 1000; run new_1024b "X=1" 1024 1000; run prev_1024b "$P" 1024 1000
run new_4096 "X=1" 4096 600; run prev_4096 "$P" 4096 600; run new_2048 "X=1" 2048 600; run prev_2048 "$P" 2048 600
} 2>&1 | tee gpurun_out/r_ab.txt
