// dsact_tu.h -- what a kernel-family translation unit of libdsact.so consists of (csrc/dsact_tu_<group>.hip): every kernel header,
// then explicit instantiation definitions of ITS group's template kernels from dsact_instances.inc. dsact_api.hip declares all of
// them `extern template`, so the 140-odd instantiations compile in eight units side by side instead of one after the other
// (__graft_entry__.build(); scripts/gen_kernel_instances.py regenerates the list). Device code is per unit (no -fgpu-rdc): a kernel
// lives in the code object of the unit that instantiates it and is launched through its host-side handle from dsact_api.hip.
#pragma once
#define DSACT_FAMILY_UNIT 1   // the non-template kernels of the headers are compiled in dsact_api.hip only
#include <hip/hip_runtime.h>

#include "dsact_kernels.h"
#include "dsact_chain.h"
#include "dsact_fat.h"
#include "dsact_act.h"
#include "dsact_conv.h"

#define DSACT_KERNEL(group, ...) DSACT_K_##group(__VA_ARGS__)
#define DSACT_INSTANTIATE(...) template __global__ void __VA_ARGS__;
#define DSACT_SKIP(...)
