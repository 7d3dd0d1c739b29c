#!/bin/bash
# round 6: row groups through the MFMA A-matrix broadcast (one LDS read per step whatever the rows per workgroup) -- tests of every form that
# mixes 4- / 8- / 16-row workgroups, then A/B against the previous library (build/libdsact_prev.so)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_groups.py tests/test_hip_cnn_parity.py tests/test_hip_v1_parity.py -q -x -p no:cacheprovider > gpurun_out/p_tests.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/p_tests.log
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps $4 --warmup 200 --batch $3 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
cnn() { echo "== $1"; env $2 timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline 2>&1 | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())['cnn']
print('   %.0f steps/s  %.1f us   %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.1f' % (k['name'], k['us']) for k in d.get('kernels', []) if 'chain' in k['name'] or k['name'] in ('dW','dfeat'))))"; }
P="DSACT_LIB_PATH=$PWD/build/libdsact_prev.so"
{
run new_256 "X=1" 256 4000; run prev_256 "$P" 256 4000; run new_256b "X=1" 256 4000; run prev_256b "$P" 256 4000
run new_512 "X=1" 512 1000; run prev_512 "$P" 512 1000
run new_1024 "X=1" 1024 1000; run prev_1024 "$P" 1024 1000
run new_4096 "X=1" 4096 600; run prev_4096 "$P" 4096 600
cnn new_cnn "X=1"; cnn prev_cnn "$P"; cnn new_cnn2 "X=1"; cnn prev_cnn2 "$P"
} 2>&1 | tee gpurun_out/p_ab.txt
