#!/bin/bash
# the CNN part of scripts/gpu_final.sh on its own (bench object, kernel stats, PMC traffic per kernel)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=$PWD/gpurun_out/final; mkdir -p $OUT
sed -n '/^# ---- CNN workload/,$p' scripts/gpu_final.sh > /tmp/cnn_part.sh
. /tmp/cnn_part.sh
