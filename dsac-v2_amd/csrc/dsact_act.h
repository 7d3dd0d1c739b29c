// dsact_act.h -- the sampler's batch-1 policy forward (training/off_sampler.py:48-51, networks/mlp.py:79-100) as ONE
// launch with no copies around it.
//
// The reference evaluates `networks.policy(obs)` on a [1, O] tensor once per environment step. Through the training
// kernels that was: pageable H2D copy of the observation, L tile stages + a head launch, D2H copy of the logits, stream
// sync -- 67 us, twenty times per iteration, against a 65 us update. Here:
//   * the observation travels INSIDE the kernel arguments (<= 768 floats), the logits come back through mapped host
//     memory, and the host spins on a mapped counter: no memcpy calls, no stream synchronisation;
//   * every layer is a block range of the same grid: wave w of layer l computes ONE output feature (its weight row is
//     read coalesced straight from the parameter arena -- always the live weights, no packed copy to keep fresh --
//     before anything waits). Layers hand their activations over as (value, call number) PAIRS written with one 8-byte
//     agent-scope store; a consumer lane spins on the pair it needs until the tag is this call's -- the data IS the flag
//     (the low-latency protocol of collective libraries): the critical path of a layer boundary is one write-through
//     plus one load, ~3 us. Consumers only wait for lower block ids, which the dispatcher has placed before them: the
//     bounded spins cannot deadlock. The logits reach the host the same way: (value, call) pairs in mapped memory.
//   Measured (profiles/r03_acting_forward.txt): separate arrival counters + data cost ~6 us per layer (store ack, atomic,
//   poll, load: 34.5 us from launch to the host seeing the result); ONE workgroup for the whole net avoids the exchange
//   but pulls 0.94 MB through a single CU with ~48 KB in flight: 40 us.
#pragma once
#include "dsact_chain.h"

namespace dsact {

constexpr int kActMaxObs = 768;
constexpr int kActMaxLayers = kChMaxL + 1;

struct ActLayer { const float* W; const float* b; int K, N; int half; };   // row-major [N][K] in the parameter arena; half > 0: a
                                                                         // twin-trunk hidden layer (policy_std_type "mlp_separated"): two [half][K]
                                                                         // blocks one after the other, rows >= half read inputs [K, 2K)
struct ActArgs {
  ActLayer ly[kActMaxLayers];
  int n_layers;                     // hidden layers + the output layer
  int wg_begin[kActMaxLayers + 1];  // block range of each layer (4 output features per workgroup)
  unsigned long long* h;            // device scratch [kActMaxLayers][kMaxWidth]: (call << 32 | float bits) per activation
  int call;                         // 1-based number of this launch
  int A; float lo_ls, hi_ls;
  int act;                          // the policy's hidden activation (ACT_*)
  unsigned long long* out;          // MAPPED HOST memory: 2A pairs (call << 32 | float bits) of (mean | std)
  int* timeout;                     // the hand-off word (mapped host memory, see check_handoff)
  // sample mode (dsact_act_sample: TanhGaussDistribution.sample() of utils/act_distribution_cls.py:32-42 in the output layer's
  // epilogue): ONE wave per action dimension d computes both of its logits (rows d and A + d), draws
  // a = scale * tanh(mean + std * eps[d]) + center and this dimension's log-prob term; out = A (action | log-prob term) pairs
  int sample;
  int out_act, out_n;               // policy_output_activation (ACT_* id, 0: linear) and the outputs it applies to (2A, or A: mean half only)
  const float* act_scale; const float* act_center;
  float eps[32];                    // the host's torch.randn(1, A) draw (consumes the generator as Normal.sample() does)
  float x[kActMaxObs];              // the observation
};
static_assert(sizeof(ActArgs) <= 4096, "kernel arguments are limited to 4 KB");

__device__ __forceinline__ unsigned long long act_pair(float v, int call) {
  return ((unsigned long long)(unsigned)call << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
}

#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(256) k_act_mlp(ActArgs a) {
  const int b = (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int l = 0;
#pragma unroll
  for (int q = 1; q < kActMaxLayers; ++q) if (q < a.n_layers && b >= a.wg_begin[q]) l = q;
  const ActLayer& Ly = a.ly[l];
  const int n = (b - a.wg_begin[l]) * 4 + wave;
  const bool smp = a.sample && l + 1 == a.n_layers;      // output layer in sample mode: waves 0 .. A-1, two rows each
  if (n >= (smp ? a.A : Ly.N)) return;        // whole wave
  // this wave's weight row, fetched before anything waits: lane k, k + 64, ...
  constexpr int NJ = (kMaxWidth > kActMaxObs ? kMaxWidth : kActMaxObs) / 64;
  float w[NJ];
  const float* wr = Ly.W + (size_t)n * Ly.K;
#pragma unroll
  for (int j = 0; j < NJ; ++j) w[j] = 64 * j + lane < Ly.K ? wr[64 * j + lane] : 0.f;
  const float bias = Ly.b[n];
  // (sample mode: the raw log-std row of the same action dimension; the output layer never reads the kernel arguments' x)
  constexpr int NJ2 = kMaxWidth / 64;
  float w2[NJ2];
  float bias2 = 0.f, s_eps = 0.f, s_scale = 1.f, s_center = 0.f;
  if (smp) {
    const float* wr2 = Ly.W + (size_t)(a.A + n) * Ly.K;
#pragma unroll
    for (int j = 0; j < NJ2; ++j) w2[j] = 64 * j + lane < Ly.K ? wr2[64 * j + lane] : 0.f;
    bias2 = Ly.b[a.A + n];
    typedef __attribute__((address_space(4))) const char KChar2;
    typedef __attribute__((address_space(4))) const float KFloat2;
    s_eps = ((KFloat2*)((KChar2*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(ActArgs, eps)))[n];
    s_scale = a.act_scale[n]; s_center = a.act_center[n];
  }
  float acc = 0.f, acc2 = 0.f;
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
    const unsigned long long* hin = a.h + (size_t)(l - 1) * kMaxWidth + ((Ly.half > 0 && n >= Ly.half) ? Ly.K : 0);   // (wave-uniform)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (64 * j < Ly.K) {                    // wave-uniform
        const int k = 64 * j + lane;
        float v = 0.f;
        if (k < Ly.K) {
          unsigned long long p;
          int spins = 0;
          for (;;) {
            p = __hip_atomic_load(hin + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(p >> 32) == a.call) break;
            if (++spins > (1 << 20)) { if (a.timeout) *a.timeout = 1; break; }   // the acting forward's OWN word (check_handoff)
          }
          v = __builtin_bit_cast(float, (unsigned)p);
        }
        acc = fmaf(w[j], v, acc);
        if (smp && j < NJ2) acc2 = fmaf(w2[j], v, acc2);
      }
    }
  }
  acc = wave_sum(acc) + bias;
  if (smp) acc2 = wave_sum(acc2) + bias2;
  if (lane != 0) return;
  if (l + 1 < a.n_layers) {
    float hv, gd;
    act_fwd_grad(a.act, acc, hv, gd);
    __hip_atomic_store(a.h + (size_t)l * kMaxWidth + n, act_pair(hv, a.call), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (a.out_act) {   // (wave-uniform) the module that follows the last Linear (networks/mlp.py:15-20)
    if (smp) { acc = out_act_fwd(a.out_act, acc); if (a.A + n < a.out_n) acc2 = out_act_fwd(a.out_act, acc2); }
    else if (n < a.out_n) acc = out_act_fwd(a.out_act, acc);
  }
  if (smp) {
    // TanhGaussDistribution.sample(): the closed form the training kernels use for rsample (dsact_math.h), term for term
    const TanhGaussFwd f = tanh_gauss_fwd(acc, acc2, s_eps, s_scale, s_center, a.lo_ls, a.hi_ls);
    __hip_atomic_store(a.out + n, act_pair(f.a, a.call), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.out + a.A + n, act_pair(f.lp, a.call), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // output layer: (mean | exp(clamp(log_std))) as StochaPolicy.forward returns them (networks/mlp.py:85-100)
  const float v = n < a.A ? acc : expf(clampf(acc, a.lo_ls, a.hi_ls));
  __hip_atomic_store(a.out + n, act_pair(v, a.call), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

}  // namespace dsact
