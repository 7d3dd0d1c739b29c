#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
run() { echo "== $1"; env $2 timeout 300 python bench.py --steps 1000 --warmup 200 --batch 1024 --no-cpu-baseline --no-alt 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('   value %.0f  us %.2f  kernels %s' % (d['value'], 1000 * d['ms_per_step'], ' '.join('%s=%.2f' % (k['name'], k['us']) for k in d.get('kernels', []))))"; }
run default "X=1"
run fat_bwd_1024 "DSACT_FAT_BWD_MIN=1024"
run no_fat "DSACT_NO_FAT=1"
run fat_rt2 "DSACT_FAT_RT=2"
run rg2 "DSACT_CHAIN_RG=2"
run default2 "X=1"
