#!/bin/bash
# builds and runs the row-slice chain microbenchmark on the GPU box
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -o /tmp/slice_gemm scripts/ubench/slice_gemm.hip || exit 1
{
for cfg in "4 16" "1 1"; do
  timeout 120 /tmp/slice_gemm $cfg
done
} 2>&1 | tee gpurun_out/ubench_slice.txt
