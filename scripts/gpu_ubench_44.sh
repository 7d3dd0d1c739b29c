#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -o /tmp/sg44 scripts/ubench/slice_gemm44.hip || exit 1
{ for cfg in "1 1" "6 32" "8 32"; do timeout 120 /tmp/sg44 $cfg; done; } 2>&1 | tee gpurun_out/ubench_slice44.txt
