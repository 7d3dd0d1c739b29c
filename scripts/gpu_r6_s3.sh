#!/bin/bash
# round 6: scheduling strategies of the whole library, alternating on one box: max-ilp (the build), the default strategy (build/libdsact_prev.so), max-memory-clause
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { echo -n "$1 "; env $2 timeout 300 python bench.py --steps $4 --warmup 200 --batch $3 --no-cpu-baseline --no-alt --headline-only 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.0f  %.2f us' % (d['value'], 1000 * d['ms_per_step']))"; }
cnn() { echo -n "$1 "; env $2 timeout 400 python bench.py --cnn-only --cnn-steps 400 --no-cpu-baseline 2>&1 | grep '^{"cnn"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())['cnn']
print('%.0f steps/s  %.1f us' % (d['value'], 1000 * d['ms_per_step']))"; }
D="DSACT_LIB_PATH=$PWD/build/libdsact_prev.so"; M="DSACT_LIB_PATH=$PWD/build/libdsact_max-memory-clause.so"
{
for i in 1 2 3 4; do run maxilp_256 "X=1" 256 4000; run default_256 "$D" 256 4000; run memclause_256 "$M" 256 4000; done
for i in 1 2 3; do run maxilp_1024 "X=1" 1024 1000; run default_1024 "$D" 1024 1000; run memclause_1024 "$M" 1024 1000; done
cnn maxilp_cnn "X=1"; cnn default_cnn "$D"; cnn memclause_cnn "$M"
} 2>&1 | tee gpurun_out/s3_ab.txt
