#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/cu_stream scripts/ubench/cu_stream.hip || exit 1
{ for nb in 1 64; do timeout 120 /tmp/cu_stream $nb; done; } 2>&1 | tee gpurun_out/ubench_cu_stream.txt
