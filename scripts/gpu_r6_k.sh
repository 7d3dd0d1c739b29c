#!/bin/bash
# round 6: start-up round trips of the chain units (late_wait as a scalar load; the critics' std sums behind the row operands) -- headline A/B
# against the previous library (build/libdsact_noprio.so), launch-form tests
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_k; rm -rf $OUT; mkdir -p $OUT
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt $3 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
{
run new "X=1" ""
run prev "DSACT_LIB_PATH=$PWD/build/libdsact_noprio.so" ""
run new2 "X=1" ""
run prev2 "DSACT_LIB_PATH=$PWD/build/libdsact_noprio.so" ""
run new_b1024 "X=1" "--batch 1024 --steps 1000 --warmup 200"
run prev_b1024 "DSACT_LIB_PATH=$PWD/build/libdsact_noprio.so" "--batch 1024 --steps 1000 --warmup 200"
} 2>&1 | tee $OUT/ab.txt
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_groups.py -q -x -p no:cacheprovider -k "pipelined or merged or graph_replay or run_group or humanoid_b256 or timeout or withhold" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
