#!/bin/bash
# round 6: headline A/B (new vs build/libdsact_prev.so) after the std-sum change + the batch legs of the evidence set on this tree
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/r6_prof; mkdir -p $OUT
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps 4000 --warmup 400 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f' % (d['value'], 1000 * d['ms_per_step']))"; }
P="DSACT_LIB_PATH=$PWD/build/libdsact_prev.so"
{ run new "X=1"; run prev "$P"; run new2 "X=1"; run prev2 "$P"; run new3 "X=1"; run prev3 "$P"; } 2>&1 | tee gpurun_out/o_ab.txt
line() { grep '^{"metric"' $1 | tail -1; }
for b in 512 1024 4096; do
  timeout 300 python bench.py --steps 1000 --warmup 200 --batch $b --no-cpu-baseline --no-alt > $OUT/bench_b$b.log 2>&1; echo "batch $b rc=$?"; line $OUT/bench_b$b.log | cut -c1-120
done
timeout 600 python bench.py --steps 1000 --warmup 200 --batch 1024 --rows 10000000 --no-cpu-baseline --no-alt > $OUT/bench_b1024_rows10M.log 2>&1; echo "10M rc=$?"; line $OUT/bench_b1024_rows10M.log | cut -c1-120
