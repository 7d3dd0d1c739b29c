#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps 1000 --warmup 200 --batch $3 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
{
run default_1024 "X=1" 1024
run dw_range_256_1024 "DSACT_DW_RANGE_256=1" 1024
run default_512 "X=1" 512
run dw_range_256_512 "DSACT_DW_RANGE_256=1" 512
run fatmin512_512 "DSACT_FAT_MIN=512" 512
} 2>&1 | tee gpurun_out/l_ab.txt
