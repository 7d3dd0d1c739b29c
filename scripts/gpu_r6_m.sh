#!/bin/bash
# round 6: the critics' std column sums with four trips' requests out before the first add -- A/B against the previous library at batch 1024 / 4096 / 256
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps $4 --warmup 200 --batch $3 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
P="DSACT_LIB_PATH=$PWD/build/libdsact_prev.so"
{
run new_1024 "X=1" 1024 1000
run prev_1024 "$P" 1024 1000
run new_1024b "X=1" 1024 1000
run prev_1024b "$P" 1024 1000
run new_4096 "X=1" 4096 600
run prev_4096 "$P" 4096 600
run new_512 "X=1" 512 1000
run prev_512 "$P" 512 1000
} 2>&1 | tee gpurun_out/m_ab.txt
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -p no:cacheprovider -k "throughput_regime or large_batch or b512 or b2048 or pipelined_graph" 2>&1 | tail -3
