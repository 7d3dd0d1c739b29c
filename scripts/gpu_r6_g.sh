#!/bin/bash
# round 6 (second session): staged input rows in the throughput-regime forward -- tests, then A/B at batch 1024 / 4096
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k "throughput_regime" 2>&1 | tail -5 > gpurun_out/g_tests.txt
cat gpurun_out/g_tests.txt
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps 1000 --warmup 200 --batch $3 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
{
run default_1024 "X=1" 1024
run global_1024 "DSACT_NO_FAT_STAGE=1" 1024
run default_1024b "X=1" 1024
run global_1024b "DSACT_NO_FAT_STAGE=1" 1024
run default_4096 "X=1" 4096
run global_4096 "DSACT_NO_FAT_STAGE=1" 4096
run default_2048 "X=1" 2048
run global_2048 "DSACT_NO_FAT_STAGE=1" 2048
} 2>&1 | tee gpurun_out/g_ab.txt
bash scripts/gpu_r6_i.sh
