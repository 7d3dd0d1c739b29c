#!/bin/bash
# same-XCD vs cross-XCD hand-over latency probe (scripts/ubench/xcd_handover.hip); usage: gpurun -- 'bash scripts/gpu_r5_ubench_handover.sh'
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/xcd_handover scripts/ubench/xcd_handover.hip || exit 1
timeout 240 /tmp/xcd_handover 2000 2>&1 | tee gpurun_out/r5_ubench_xcd_handover.txt
