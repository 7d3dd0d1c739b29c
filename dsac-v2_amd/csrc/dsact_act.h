// dsact_act.h -- the sampler's batch-1 policy forward (training/off_sampler.py:48-51, networks/mlp.py:79-100) as ONE
// launch with no copies around it.
//
// The reference evaluates `networks.policy(obs)` on a [1, O] tensor once per environment step. Through the training
// kernels that was: pageable H2D copy of the observation, L tile stages + a head launch, D2H copy of the logits, stream
// sync -- 67 us, twenty times per iteration, against a 65 us update. Here:
//   * the observation travels INSIDE the kernel arguments (<= 768 floats), the logits come back through mapped host
//     memory, and the host spins on a mapped counter: no memcpy calls, no stream synchronisation;
//   * every layer is a block range of the same grid: wave w of layer l computes ONE output feature (its weight row is
//     read coalesced straight from the parameter arena -- always the live weights, no packed copy to keep fresh) and a
//     layer's workgroups start multiplying when the previous layer's arrival counter is complete. Consumers only wait
//     for lower block ids, which the dispatcher has placed before them: the bounded spins cannot deadlock. Hand-over
//     data (<= 4 KB per layer) is written and read at agent scope, like the merged forward of dsact_chain.h.
#pragma once
#include "dsact_chain.h"

namespace dsact {

constexpr int kActMaxObs = 768;
constexpr int kActMaxLayers = kChMaxL + 1;

struct ActLayer { const float* W; const float* b; int K, N; };   // row-major [N][K] in the parameter arena
struct ActArgs {
  ActLayer ly[kActMaxLayers];
  int n_layers;                     // hidden layers + the output layer
  int wg_begin[kActMaxLayers + 1];  // block range of each layer (4 output features per workgroup)
  float* h[2];                      // device scratch, ping-pong: activations of the even / odd layers
  int* cnt;                         // device arrival counters, one per layer, monotone: complete at call * workgroups
  int call;                         // 1-based number of this launch
  int A; float lo_ls, hi_ls;
  float* out;                       // MAPPED HOST memory: (mean | std), 2A floats
  int* done;                        // MAPPED HOST memory: monotone counter of finished output workgroups
  int* timeout;                     // the hand-off word (mapped host memory, see check_handoff)
  float x[kActMaxObs];              // the observation
};
static_assert(sizeof(ActArgs) <= 4096, "kernel arguments are limited to 4 KB");

__global__ void __launch_bounds__(256) k_act_mlp(ActArgs a) {
  const int b = (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int l = 0;
#pragma unroll
  for (int q = 1; q < kActMaxLayers; ++q) if (q < a.n_layers && b >= a.wg_begin[q]) l = q;
  const ActLayer& Ly = a.ly[l];
  const int n = (b - a.wg_begin[l]) * 4 + wave;
  const bool live = n < Ly.N;
  // this wave's weight row, fetched before anything waits: lane k, k + 64, ...
  constexpr int NJ = (kMaxWidth > kActMaxObs ? kMaxWidth : kActMaxObs) / 64;
  float w[NJ];
  const float* wr = Ly.W + (size_t)(live ? n : 0) * Ly.K;
#pragma unroll
  for (int j = 0; j < NJ; ++j) w[j] = 64 * j + lane < Ly.K ? wr[64 * j + lane] : 0.f;
  const float bias = live ? Ly.b[n] : 0.f;
  float acc = 0.f;
  if (l == 0) {
    // the observation sits in the kernel-argument segment: read it as memory (indexing the by-value struct with a
    // lane-dependent index would make the compiler spill the whole 4 KB argument to scratch)
    typedef __attribute__((address_space(4))) const char KChar;
    typedef __attribute__((address_space(4))) const float KFloat;
    KFloat* xk = (KFloat*)((KChar*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ActArgs, x));
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (64 * j + lane < Ly.K) acc = fmaf(w[j], xk[64 * j + lane], acc);
  } else {
    const int need = a.call * (a.wg_begin[l] - a.wg_begin[l - 1]);
    if (tid == 0) {
      int spins = 0;
      while (__hip_atomic_load(a.cnt + (l - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - need < 0) {
        if (++spins > (1 << 20)) { if (a.timeout) *a.timeout = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    __syncthreads();
    const float* hin = a.h[(l - 1) & 1];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (64 * j + lane < Ly.K) acc = fmaf(w[j], ld_agent(hin + 64 * j + lane), acc);
  }
  acc = wave_sum(acc) + bias;
  if (l + 1 < a.n_layers) {
    float hv, gd;
    gelu_fwd_grad(acc, hv, gd);
    if (live && lane == 0) st_agent(a.h[l & 1] + n, hv);
    stores_acked_barrier();
    if (tid == 0) __hip_atomic_fetch_add(a.cnt + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // output layer: (mean | exp(clamp(log_std))) as StochaPolicy.forward returns them (networks/mlp.py:85-100)
  if (live && lane == 0) {
    const float v = n < a.A ? acc : expf(clampf(acc, a.lo_ls, a.hi_ls));
    __hip_atomic_store(a.out + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: the logits are in host memory before the counter moves
  stores_acked_barrier();
  if (tid == 0) __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace dsact
