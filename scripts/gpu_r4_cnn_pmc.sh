#!/bin/bash
# PMC passes of the CNN workload only (the tail of scripts/gpu_final.sh without the benches)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/final; mkdir -p $OUT
sed -n '/^for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do$/,$p' scripts/gpu_final.sh | awk 'BEGIN{n=0} /^for pmc in FETCH_SIZE WRITE_SIZE MfmaUtil; do$/{n++} n>=2{print}' | sed '/^timeout 300 python bench.py --steps 2000/d' > /tmp/cnn_pmc_part.sh
. /tmp/cnn_pmc_part.sh
