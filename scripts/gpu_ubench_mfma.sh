#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -save-temps=obj -o /tmp/mfma_operands scripts/ubench/mfma_operands.hip || exit 1
/tmp/mfma_operands 2>&1 | tee gpurun_out/ubench_mfma_operands.txt
grep -c v_mfma /tmp/mfma_operands-hip-amdgcn-amd-amdhsa-gfx950.s
awk '/^_Z1kILi16ELi16ELi4EE/,/s_endpgm/' /tmp/mfma_operands-hip-amdgcn-amd-amdhsa-gfx950.s | grep -A40 "s_memtime\|s_getreg\|readcyclecounter\|BB" | grep -v "^--" | head -60 > gpurun_out/ubench_mfma_isa.txt
