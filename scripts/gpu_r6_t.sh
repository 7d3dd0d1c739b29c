#!/bin/bash
# round 6: kernel-argument preload into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16; build/libdsact_xnackoff.so) against the build, alternating
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() { echo -n "$1 "; env $2 timeout 300 python bench.py --steps $4 --warmup 200 --batch $3 --no-cpu-baseline --no-alt --headline-only 2>&1 | grep '^{"metric"' | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.0f  %.2f us' % (d['value'], 1000 * d['ms_per_step']))"; }
K="DSACT_LIB_PATH=$PWD/build/libdsact_xnackoff.so"
{
for i in 1 2 3 4; do run build_256 "X=1" 256 4000; run xnackoff_256 "$K" 256 4000; done
for i in 1 2; do run build_1024 "X=1" 1024 1000; run xnackoff_1024 "$K" 1024 1000; done
} 2>&1 | tee gpurun_out/t_ab.txt
