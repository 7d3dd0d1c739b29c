// dsact_act.h -- the sampler's batch-1 policy forward (training/off_sampler.py:48-51, networks/mlp.py:79-100) as ONE
// launch with no copies around it.
//
// The reference evaluates `networks.policy(obs)` on a [1, O] tensor once per environment step. Through the training
// kernels that was: pageable H2D copy of the observation, L tile stages + a head launch, D2H copy of the logits, stream
// sync -- 67 us, twenty times per iteration, against a 65 us update. Here:
//   * the observation travels INSIDE the kernel arguments (<= 768 floats), the logits come back through mapped host
//     memory, and the host spins on a mapped counter: no memcpy calls, no stream synchronisation;
//   * ONE workgroup of 16 waves runs the whole net: wave w computes output features w, w + 16, ... of a layer (its
//     weight rows are read coalesced straight from the parameter arena -- always the live weights, no packed copy to
//     keep fresh), the activations are handed from layer to layer through LDS. 0.94 MB of weights through one CU's L2
//     port is ~6 us; the first version spread the layers over 200 workgroups with arrival counters, and paid ~6 us PER
//     LAYER for the agent-scope (cross-XCD) store / counter / load round trips (45 us per call end to end).
#pragma once
#include "dsact_chain.h"

namespace dsact {

constexpr int kActMaxObs = 768;
constexpr int kActMaxLayers = kChMaxL + 1;

struct ActLayer { const float* W; const float* b; int K, N; };   // row-major [N][K] in the parameter arena
struct ActArgs {
  ActLayer ly[kActMaxLayers];
  int n_layers;                     // hidden layers + the output layer
  int call;                         // 1-based number of this launch
  int A; float lo_ls, hi_ls;
  float* out;                       // MAPPED HOST memory: (mean | std), 2A floats
  int* done;                        // MAPPED HOST memory: set to `call` when the logits are there
  float x[kActMaxObs];              // the observation
};
static_assert(sizeof(ActArgs) <= 4096, "kernel arguments are limited to 4 KB");

constexpr int kActWaves = 16;
__global__ void __launch_bounds__(64 * kActWaves) k_act_mlp(ActArgs a) {
  __shared__ float hbuf[2][kMaxWidth];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the observation sits in the kernel-argument segment: read it as memory (indexing the by-value struct with a
  // lane-dependent index would make the compiler spill the whole 4 KB argument to scratch)
  typedef __attribute__((address_space(4))) const char KChar;
  typedef __attribute__((address_space(4))) const float KFloat;
  KFloat* xk = (KFloat*)((KChar*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ActArgs, x));
  for (int k = tid; k < a.ly[0].K; k += 64 * kActWaves) hbuf[1][k] = xk[k];   // layer l reads hbuf[(l + 1) & 1]
  __syncthreads();
  constexpr int NJ = (kMaxWidth > kActMaxObs ? kMaxWidth : kActMaxObs) / 64;
  for (int l = 0; l < a.n_layers; ++l) {
    const ActLayer Ly = a.ly[l];
    const float* in = hbuf[(l + 1) & 1];
    float* outb = hbuf[l & 1];
    const bool last = l + 1 == a.n_layers;
    float xin[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) xin[j] = 64 * j + lane < Ly.K ? in[64 * j + lane] : 0.f;
    // two output features per trip: their weight rows are in flight together
    for (int n0 = wave; n0 < Ly.N; n0 += 2 * kActWaves) {
      const int n1 = n0 + kActWaves;
      const bool v1 = n1 < Ly.N;
      const float* w0 = Ly.W + (size_t)n0 * Ly.K;
      const float* w1 = Ly.W + (size_t)(v1 ? n1 : n0) * Ly.K;
      float r0[NJ], r1[NJ];
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const bool kv = 64 * j < Ly.K;      // wave-uniform: whole trips beyond K are skipped
        r0[j] = kv && 64 * j + lane < Ly.K ? w0[64 * j + lane] : 0.f;
        r1[j] = kv && 64 * j + lane < Ly.K ? w1[64 * j + lane] : 0.f;
      }
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j) { a0 = fmaf(r0[j], xin[j], a0); a1 = fmaf(r1[j], xin[j], a1); }
      a0 = wave_sum(a0) + Ly.b[n0];
      a1 = wave_sum(a1) + (v1 ? Ly.b[n1] : 0.f);
      if (lane == 0) {
        if (!last) {
          float hv, gd;
          gelu_fwd_grad(a0, hv, gd); outb[n0] = hv;
          if (v1) { gelu_fwd_grad(a1, hv, gd); outb[n1] = hv; }
        } else {
          // output layer: (mean | exp(clamp(log_std))) as StochaPolicy.forward returns them (networks/mlp.py:85-100)
          __hip_atomic_store(a.out + n0, n0 < a.A ? a0 : expf(clampf(a0, a.lo_ls, a.hi_ls)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (v1) __hip_atomic_store(a.out + n1, n1 < a.A ? a1 : expf(clampf(a1, a.lo_ls, a.hi_ls)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (!last) __syncthreads();
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the logits are in host memory before the word moves
  stores_acked_barrier();
  if (tid == 0) __hip_atomic_store(a.done, a.call, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace dsact
