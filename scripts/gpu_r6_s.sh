#!/bin/bash
# round 6: the library compiled with the MaxILP scheduling strategy (build/libdsact_maxilp.so) against the default -- headline, batch 1024, CNN
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps $4 --warmup 200 --batch $3 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f' % (d['value'], 1000 * d['ms_per_step']))"; }
cnn() { echo "== $1"; env $2 timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline 2>&1 | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())['cnn']
print('   %.0f steps/s  %.1f us' % (d['value'], 1000 * d['ms_per_step']))"; }
P="DSACT_LIB_PATH=$PWD/build/libdsact_maxilp.so"
{
run default_256 "X=1" 256 4000; run maxilp_256 "$P" 256 4000; run default_256b "X=1" 256 4000; run maxilp_256b "$P" 256 4000
run default_1024 "X=1" 1024 1000; run maxilp_1024 "$P" 1024 1000
cnn default_cnn "X=1"; cnn maxilp_cnn "$P"
} 2>&1 | tee gpurun_out/s_ab.txt
