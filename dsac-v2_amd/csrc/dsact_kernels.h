// dsact_kernels.h -- gfx950 (MI355X / CDNA4) kernels of the DSAC-T update. HIP only, wave64.
//
// One update (= DSAC_V2.local_update, dsac_v2.py:102-105) is a short chain of launches:
//   k_gather   replay rows -> minibatch staging (+ per-step bookkeeping, device RNG)
//   k_tiles    task-table driven 32x32 output tiles on the fp32 matrix cores
//              (v_mfma_f32_16x16x4_f32: bit-exact fmaf chain, same peak as the fp32 VALU),
//              LDS-staged operands, fused bias+GELU / GELU' / weight-gradient epilogues
//   k_heads    output layers + tanh-Gaussian rsample (wave per row, shuffle reductions)
//   k_loss     twin distributional-Q target with the three DSAC-T refinements, actor/alpha loss,
//              output-layer gradients, last-hidden-layer dZ
//   k_heads_bwd  dQ/da -> rsample backward -> policy output-layer gradient
//   k_adam     fused Adam (q1,q2 every step; policy, log_alpha delayed) + Polyak target sync
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dsact_math.h"

namespace dsact {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// 4-byte aligned 16-byte vector: gfx950 global memory runs in unaligned-access mode, so a
// dword-aligned dwordx4 is legal (weight rows of the Q nets have odd leading dimension O+A).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// 4-wide device form of gelu_fwd_grad (dsact_math.h): the same FMA sequence per element, written on vectors so that
// the polynomial chains compile to packed v_pk_fma_f32 (two elements per instruction)
__device__ __forceinline__ f32x4 vfma4(f32x4 a, f32x4 b, f32x4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x4 splat4(float v) { return f32x4{v, v, v, v}; }
// Activations / gradients written for the NEXT launch (read by other XCDs, never again by this one): streaming stores
// leave the L2 as they are issued instead of in the kernel-end write-back.
#ifndef DSACT_NT_STORES
#define DSACT_NT_STORES 1
#endif
__device__ __forceinline__ void nt_store4(float* p, f32x4 v) {
#if DSACT_NT_STORES
  __builtin_nontemporal_store(v, (f32x4*)p);
#else
  *(f32x4*)p = v;
#endif
}
__device__ __forceinline__ void gelu4(const f32x4 z, f32x4& h, f32x4& g) {
  const f32x4 x = z * kInvSqrt2;
  const f32x4 t = __builtin_elementwise_min(__builtin_elementwise_abs(x), splat4(kErfHi));
  const f32x4 s = x * x;
  f32x4 rs = splat4(kErfS[5]);
#pragma unroll
  for (int k = 4; k >= 0; --k) rs = vfma4(rs, s, splat4(kErfS[k]));
  const f32x4 small = vfma4(rs, x, x);
  f32x4 rl = splat4(kErfL[8]);
#pragma unroll
  for (int k = 7; k >= 0; --k) rl = vfma4(rl, t, splat4(kErfL[k]));
  const f32x4 arg = vfma4(rl, t, -t) * kLog2e;
  const f32x4 parg = (z * -0.5f) * z * kLog2e;
  f32x4 erf, pdf;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float large = copysignf(1.0f - __builtin_amdgcn_exp2f(arg[e]), x[e]);
    erf[e] = t[e] > kErfT0 ? large : small[e];
    pdf[e] = kInvSqrt2Pi * __builtin_amdgcn_exp2f(parg[e]);
  }
  const f32x4 cdf = vfma4(erf, splat4(0.5f), splat4(0.5f));
  h = z * cdf;
  g = vfma4(z, pdf, cdf);
}

// act(z), act'(z) on 4 values; `act` is wave-uniform (a net's hidden activation): GELU takes the packed path above
__device__ __forceinline__ void act4(int act, const f32x4 z, f32x4& h, f32x4& g) {
  if (act == ACT_GELU) { gelu4(z, h, g); return; }
#pragma unroll
  for (int e = 0; e < 4; ++e) { float hv, gv; act_fwd_grad(act, z[e], hv, gv); h[e] = hv; g[e] = gv; }
}

// ---- fragment-major ("packed") copies of a row-major [N x K] matrix -------------------------------------------
// style 16 (v_mfma_f32_16x16x4_f32 operands; the narrow products: output layers, dL/d action):
//   tiles of 16 rows x chunks of 16 k; inside a (tile, chunk) block lane (g = (k%16)/4, i = n%16) owns the 4 floats
//   k%4 = 0..3 -- the operand of 4 consecutive MFMA steps (k-slot trick below). C = chunks per tile.
// style 44 (v_mfma_f32_4x4x1_16b_f32 operands; every full-width layer): lane <-> row n of a 64-row tile, one wave-load
//   = the 64 rows' 4 consecutive k = 1 KB contiguous: [n/64][k/4][n%64][k%4]. C = k4 steps per tile.
__host__ __device__ inline size_t pk_index(int n, int k, int C) {
  return (((size_t)(n >> 4) * C + (k >> 4)) * 64 + (size_t)((((k & 15) >> 2) << 4) + (n & 15))) * 4 + (k & 3);
}
__host__ __device__ inline size_t pk44_index(int n, int k, int C) {
  return (((size_t)(n >> 6) * C + (k >> 2)) * 64 + (size_t)(n & 63)) * 4 + (k & 3);
}

// What the owner of a weight tensor keeps fresh besides the arena (single-GPU fused optimiser: the dW/Adam tile;
// every other flow: k_pack at the start of the step).
struct MirrorDesc {
  float* fwd;      // pack of W [N x K'] (forward chains); K' = k for k < F, Fp + (k - F) beyond (first layer of a Q net:
  float* fwd_t;    //   observation columns padded to whole step groups, action columns behind); fwd_t: target net's copy
  int fwd_C, fwd_44;   // chunks (style 16) or k4 steps (style 44) per tile; fwd_44: 1 = style 44
  int F, Fp;       // F >= K: identity
  float* bwd;      // pack of (W[:, bwd_k0:])^T  [K - bwd_k0 x N] (backward chains); nullptr: none
  int bwd_C, bwd_44, bwd_k0;
};

// one lane's 4 consecutive-k values of row n (k % 4 == 0, all inside one segment) -> the copies
// A Q net's first layer whose observation width F is no multiple of 4: the source quad (k .. k+3) straddles the observation /
// action boundary or lies behind it, where the packed position Fp + (k - F) is no multiple of 4 either -- element by element
// (the action columns: at most 32 + 3 per row; every other quad keeps its 16-byte store).
__device__ __forceinline__ void mirror_fwd_each(const MirrorDesc& m, int n, int k, int K, const f32x4& v, bool target, const f32x4& vt) {
  for (int e = 0; e < 4 && k + e < K; ++e) {
    const int ke = k + e, kk = ke < m.F ? ke : m.Fp + (ke - m.F);
    const size_t o = m.fwd_44 ? pk44_index(n, kk, m.fwd_C) : pk_index(n, kk, m.fwd_C);
    m.fwd[o] = v[e];
    if (target && m.fwd_t) m.fwd_t[o] = vt[e];
  }
}
__device__ __forceinline__ void mirror_store4(const MirrorDesc& m, int n, int k, int K, const f32x4& v, bool target, const f32x4& vt) {
  if (m.fwd) {
    const int kk = k < m.F ? k : m.Fp + (k - m.F);
    const size_t o = m.fwd_44 ? pk44_index(n, kk, m.fwd_C) : pk_index(n, kk, m.fwd_C);
    if ((m.F & 3) && k + 3 >= m.F && m.F < (1 << 30)) mirror_fwd_each(m, n, k, K, v, target, vt);   // (observation width no multiple of 4)
    else if (k + 3 < K) {
      *(f32x4*)(m.fwd + o) = v;
      if (target && m.fwd_t) *(f32x4*)(m.fwd_t + o) = vt;
    } else {
      for (int e = 0; e < 4 && k + e < K; ++e) { m.fwd[o + e] = v[e]; if (target && m.fwd_t) m.fwd_t[o + e] = vt[e]; }
    }
  }
  if (m.bwd) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = k + e - m.bwd_k0;
      if (r >= 0 && k + e < K) m.bwd[m.bwd_44 ? pk44_index(r, n, m.bwd_C) : pk_index(r, n, m.bwd_C)] = v[e];
    }
  }
}

// 4x4 transpose inside a lane quad: lane j (= lane & 3) of the quad holds v_j; returns (v_0[j], v_1[j], v_2[j], v_3[j]).
template <int CTRL>
__device__ __forceinline__ float quad_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sel4(const f32x4& v, int i) { return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3]; }
__device__ __forceinline__ f32x4 quad_transpose(const f32x4& v, int q /* lane & 3 */) {
  // rotation r: lane j reads from lane c = (j + r) & 3 the component c exposes for it, v_c[(c - r) & 3] = v_c[j]
  const float t0 = sel4(v, q);
  const float t1 = quad_mov<0x39>(sel4(v, (q - 1) & 3));   // quad_perm [1,2,3,0]
  const float t2 = quad_mov<0x4E>(sel4(v, (q - 2) & 3));   // quad_perm [2,3,0,1]
  const float t3 = quad_mov<0x93>(sel4(v, (q - 3) & 3));   // quad_perm [3,0,1,2]
  f32x4 o;   // t_r belongs at component (q + r) & 3
  o[0] = q == 0 ? t0 : q == 3 ? t1 : q == 2 ? t2 : t3;
  o[1] = q == 1 ? t0 : q == 0 ? t1 : q == 3 ? t2 : t3;
  o[2] = q == 2 ? t0 : q == 1 ? t1 : q == 0 ? t2 : t3;
  o[3] = q == 3 ? t0 : q == 2 ? t1 : q == 1 ? t2 : t3;
  return o;
}
// mirror_store4 for callers whose lane quads hold 4 CONSECUTIVE rows n (n % 4 == lane % 4) at the same k: the transposed
// copy gets one 16-byte store per lane (row k + lane%4 of W^T, columns n&~3 .. +3) instead of four scattered 4-byte ones.
// Every lane of the quad must call it (DPP); `valid`: this lane's row exists (v must be zero otherwise).
__device__ __forceinline__ void mirror_store4_quad(const MirrorDesc& m, int n, int k, int K, int N, bool valid, const f32x4& v,
                                                   bool target, const f32x4& vt, int lane) {
  if (m.fwd && valid) {
    const int kk = k < m.F ? k : m.Fp + (k - m.F);
    const size_t o = m.fwd_44 ? pk44_index(n, kk, m.fwd_C) : pk_index(n, kk, m.fwd_C);
    if ((m.F & 3) && k + 3 >= m.F && m.F < (1 << 30)) mirror_fwd_each(m, n, k, K, v, target, vt);   // (observation width no multiple of 4)
    else if (k + 3 < K) {
      *(f32x4*)(m.fwd + o) = v;
      if (target && m.fwd_t) *(f32x4*)(m.fwd_t + o) = vt;
    } else {
      for (int e = 0; e < 4 && k + e < K; ++e) { m.fwd[o + e] = v[e]; if (target && m.fwd_t) m.fwd_t[o + e] = vt[e]; }
    }
  }
  if (m.bwd) {
    const int q = lane & 3;
    const f32x4 t = quad_transpose(v, q);
    const int kr = k + q, r = kr - m.bwd_k0, n4 = n & ~3;
    if (r >= 0 && kr < K && n4 < N) *(f32x4*)(m.bwd + (m.bwd_44 ? pk44_index(r, n4, m.bwd_C) : pk_index(r, n4, m.bwd_C))) = t;
  }
}

// ---- k_pack: rebuild every packed copy from the arenas (eager flows, start of a graph launch, external writes) ----
struct PackJob {
  const float* src; int N, K;   // row-major source (arena)
  MirrorDesc m;                 // destinations (fwd_t unused: the target nets are jobs of their own)
  int block_end;                // exclusive end of this job's block range (one block per 16 source rows)
  int is_target;                // a target net: unchanged on the off iterations of the delayed update
};
// targets_if: nullptr, or DevState::do_delayed -- skip the target nets' jobs when the update in flight left them alone
__device__ __forceinline__ void pack_block(const PackJob* jobs, int n_jobs, int b, int tid, const int* targets_if = nullptr) {
  int ji = 0;
  for (int q = 0; q + 1 < n_jobs; ++q) if (b >= jobs[q].block_end) ji = q + 1;
  const PackJob J = jobs[ji];
  if (J.is_target && targets_if && !*targets_if) return;
  const int n0 = (b - (ji ? jobs[ji - 1].block_end : 0)) * 16;
  const int kq = (J.K + 3) >> 2;
  // 16 consecutive lanes = the block's 16 rows at one k group: a lane quad holds 4 consecutive rows (mirror_store4_quad)
  for (int e0 = 0; e0 < 16 * kq; e0 += 256) {
    const int e = e0 + tid;
    const int n = n0 + (e & 15), k = (e >> 4) * 4;
    const bool valid = e < 16 * kq && n < J.N;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      const float* s = J.src + (size_t)n * J.K + k;
      if (k + 3 < J.K) v = *(const f32x4u*)s;
      else for (int c = 0; c < 4 && k + c < J.K; ++c) v[c] = s[c];
    }
    mirror_store4_quad(J.m, n, e < 16 * kq ? k : J.K, J.K, J.N, valid, v, false, v, tid);
  }
}
struct PackArgs { const PackJob* jobs; int n_jobs; const int* targets_if; };
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(256) k_pack(PackArgs a) { pack_block(a.jobs, a.n_jobs, (int)blockIdx.x, threadIdx.x, a.targets_if); }
#endif

constexpr int kWave = 64;
constexpr int kThreads = 256;  // 4 waves, one per SIMD
constexpr int kMaxWidth = 1024;

// ---------------------------------------------------------------------------------------------
// device-resident step state (nothing here is read back by the host on the hot path)
// ---------------------------------------------------------------------------------------------
struct DevState {
  long long it_next;   // iteration the next graph-replayed step will use (written by k_adam)
  long long it_cur;    // iteration of the step in flight (written by the prologue)
  long long seq_next;  // replayed-step sequence number -> row of the index table
  int t_q, t_pi, t_alpha;  // Adam step counters (torch: state["step"])
  int ms_init;             // 0 <=> reference sentinel mean_std == -1.0
  float ms1, ms2;          // mean_std1/2 EMA (dsac_v2.py:233-241)
  float ss_q, bc2_q;       // Adam scalars of the step in flight: lr/(1-b1^t), sqrt(1-b2^t)
  float ss_pi, bc2_pi;
  float ss_alpha, bc2_alpha;
  int do_delayed;          // it % delay_update == 0
  int pad;
  // beta1^t, beta2^t per optimiser as running products (double): device pow() is ~1 us per call
  double b1p_q, b2p_q, b1p_pi, b2p_pi, b1p_alpha, b2p_alpha;
  long long tag_seq;   // closed updates since the handle was created, NEVER reset: the (value, tag) hand-overs tag with its low
                       // word + 1 (seq_next restarts at 0 with every dsact_run_group: a tag derived from it would match the
                       // pairs a previous group left in the buffers)
};

// opt-in phase timeline (build with -DDSACT_TIMELINE): shader-clock stamps of selected blocks
#ifdef DSACT_TIMELINE
#define TL_DECL long long tl_t[8]; int tl_n = 0; const long long tl_w0 = (long long)wall_clock64();
#define TL_STAMP() do { if (tl_n < 8) tl_t[tl_n++] = (long long)__builtin_readcyclecounter(); } while (0)
#define TL_FLUSH(buf, slot) do { if ((buf) && threadIdx.x == 0 && (slot) < 512) { \
    for (int q_ = 0; q_ < 8; ++q_) (buf)[(slot) * 8 + q_] = q_ < tl_n ? tl_t[q_] : 0; \
    (buf)[(slot) * 8 + 6] = tl_w0; (buf)[(slot) * 8 + 7] = (long long)wall_clock64(); } } while (0)
#else
#define TL_DECL
#define TL_STAMP() do {} while (0)
#define TL_FLUSH(buf, slot) do {} while (0)
#endif

// Wave64 all-reduce in pure VALU: 4 DPP steps give every lane its 16-lane row total, then
// v_permlane16_swap / v_permlane32_swap (gfx950) exchange rows and halves. A dependent chain of
// __shfl_xor (ds_bpermute, an LDS-crossbar round trip per step) costs ~100+ cycles per step; this costs
// a handful. NOTE the empty asm on the swap results: with identical operands hipcc (ROCm 7.2)
// copy-propagates the two outputs into one register (emits v_add v1,v1,v1) without it.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ void swap16(float v, float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  unsigned r0 = r[0], r1 = r[1];
  asm volatile("" : "+v"(r0), "+v"(r1));
  a = __builtin_bit_cast(float, r0); b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void swap32(float v, float& a, float& b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
  unsigned r0 = r[0], r1 = r[1];
  asm volatile("" : "+v"(r0), "+v"(r1));
  a = __builtin_bit_cast(float, r0); b = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  float a, b;
  swap16(v, a, b); v = a + b;
  swap32(v, a, b); v = a + b;
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
  v = fminf(v, dpp_mov<0xB1>(v));
  v = fminf(v, dpp_mov<0x4E>(v));
  v = fminf(v, dpp_mov<0x141>(v));
  v = fminf(v, dpp_mov<0x140>(v));
  float a, b;
  swap16(v, a, b); v = fminf(a, b);
  swap32(v, a, b); v = fminf(a, b);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller (production-mode noise; parity mode injects torch.randn draws)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// 4 standard normals for (stream, index/4) of iteration `it`
__device__ __forceinline__ void normal4(uint64_t seed, long long it, uint32_t stream, uint32_t idx4, float z[4]) {
  uint32_t r[4];
  philox4x32(idx4, (uint32_t)it, (uint32_t)((uint64_t)it >> 32), stream, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float u0 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u1 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u3 = ((float)(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float ra = sqrtf(-2.0f * logf(u0)), rb = sqrtf(-2.0f * logf(u2));
  float s, c;
  sincosf(6.28318530717958647692f * u1, &s, &c);
  z[0] = ra * c; z[1] = ra * s;
  sincosf(6.28318530717958647692f * u3, &s, &c);
  z[2] = rb * c; z[3] = rb * s;
}

// ---------------------------------------------------------------------------------------------
// k_gather: replay rows -> staging (training/replay_buffer.py:85-90 + trainer.py:72-74)
//   X0 = [obs | act | 0pad], XP = [obs | (new_act later) ], X2 = [obs2 | (act2 later)], rew, done
//   one wave per row; rows are 4*O bytes contiguous in the ring (coalesced dwordx4 when O%4==0)
// block 0 additionally performs the per-step bookkeeping (see prologue_duties).
// ---------------------------------------------------------------------------------------------
struct StepHyper {
  int delay_update;
  double lr_q, lr_pi, lr_alpha, beta1, beta2;  // the decimal values of the config (Python doubles)
};

__device__ void prologue_duties(DevState* st, long long it, int advance_counters, StepHyper hp) {
  st->it_cur = it;
  const int delayed = (it % hp.delay_update) == 0;
  st->do_delayed = delayed;
  if (advance_counters) {
    // torch Adam: bias_correction1 = 1 - beta1**step ; step_size = lr/bias_correction1 ;
    //             bias_correction2_sqrt = (1 - beta2**step)**0.5      (Python doubles)
    const double b1 = hp.beta1, b2 = hp.beta2;
    st->t_q += 1;
    st->b1p_q *= b1; st->b2p_q *= b2;
    st->ss_q = (float)(hp.lr_q / (1.0 - st->b1p_q));
    st->bc2_q = (float)sqrt(1.0 - st->b2p_q);
    if (delayed) {
      st->t_pi += 1; st->t_alpha += 1;
      st->b1p_pi *= b1; st->b2p_pi *= b2;
      st->b1p_alpha *= b1; st->b2p_alpha *= b2;
      st->ss_pi = (float)(hp.lr_pi / (1.0 - st->b1p_pi));
      st->bc2_pi = (float)sqrt(1.0 - st->b2p_pi);
      st->ss_alpha = (float)(hp.lr_alpha / (1.0 - st->b1p_alpha));
      st->bc2_alpha = (float)sqrt(1.0 - st->b2p_alpha);
    }
  }
}

struct NoiseArgs {
  uint64_t seed;  // 0: noise buffers were filled by the host (parity mode)
  float* eps_new; float* eps_2; float* z5; float* z6;
  // noise TABLE (strict RNG through graph replays): row `trow` of [rows][2*B*A + 2*B] floats = eps_new | eps_2 | z5 | z6 of the
  // update that reads index-table row `trow` -- the reference's own torch.randn draws, uploaded with
  // the index rows by dsact_run_group. nullptr: device Philox keyed by (seed, iteration)
  const float* table; int B;
};

// fills the noise of rows [r0, r1) of the minibatch of iteration `it` (index / noise table row `trow`)
__device__ void fill_noise_rows(const NoiseArgs& nz, long long it, int trow, int r0, int r1, int A, int tid, int nthreads) {
  if (nz.table) {
    const size_t BA = (size_t)nz.B * A;
    const float* row = nz.table + (size_t)trow * (2 * BA + 2 * (size_t)nz.B);
    for (int q = r0 * A + tid; q < r1 * A; q += nthreads) { nz.eps_new[q] = row[q]; nz.eps_2[q] = row[BA + q]; }
    for (int r = r0 + tid; r < r1; r += nthreads) { nz.z5[r] = row[2 * BA + r]; nz.z6[r] = row[2 * BA + nz.B + r]; }
    return;
  }
  const int per_row4 = (A + 3) >> 2;
  const int n4 = (r1 - r0) * per_row4;
  for (int q = tid; q < n4; q += nthreads) {
    const int r = r0 + q / per_row4, j4 = (q % per_row4) * 4;
    float z[4];
    normal4(nz.seed, it, 1u, (uint32_t)(r * per_row4 + j4 / 4), z);
    for (int e = 0; e < 4; ++e) if (j4 + e < A) nz.eps_new[(size_t)r * A + j4 + e] = z[e];
    normal4(nz.seed, it, 2u, (uint32_t)(r * per_row4 + j4 / 4), z);
    for (int e = 0; e < 4; ++e) if (j4 + e < A) nz.eps_2[(size_t)r * A + j4 + e] = z[e];
  }
  for (int r = r0 + tid; r < r1; r += nthreads) {
    float z[4];
    normal4(nz.seed, it, 3u, (uint32_t)r, z);
    nz.z5[r] = z[0];
    nz.z6[r] = z[1];
  }
}

// zero-padded copies of the Q nets' first-layer weights: rows of O+A floats (odd leading dimension)
// become rows of `ldp` floats (multiple of 4) so that the K=O+A tile stages issue aligned dwordx4.
// Rebuilt at the start of every step (after Adam / Polyak / any external state_dict load) by spare
// blocks of the gather launch; 4 nets x W0 x ldp floats.
struct RepackArgs {
  const float* src[4]; float* dst[4];
  int rows, K, ldp;
  int n_blocks;        // blocks assigned to the repack (0 = none)
  float* w1at[2];      // action columns of q1 / q2's first layer, TRANSPOSED [32 (j, zero padded)][rows]
  int O, A;            //   (k_heads_bwd forms dL/d new_act from them with coalesced row loads)
  int skip_pad;        // 1: wide first layers (e.g. 3136 conv features): no padded copies, the stages read the arena rows
  const PackJob* pk_jobs; int pk_n_jobs;   // chain mode (dsact_chain.h): the blocks rebuild the fragment-major copies instead
};
__device__ void repack_rows(const RepackArgs& rp, int blk, int tid) {
  if (rp.pk_jobs) { pack_block(rp.pk_jobs, rp.pk_n_jobs, blk, tid); return; }
  const int per_net = rp.rows * rp.ldp;
  const int total = rp.skip_pad ? 0 : 4 * per_net;
  for (int e = blk * kThreads + tid; e < total; e += rp.n_blocks * kThreads) {
    const int net = e / per_net, rem = e - net * per_net;
    const int row = rem / rp.ldp, col = rem - row * rp.ldp;
    rp.dst[net][rem] = col < rp.K ? rp.src[net][(size_t)row * rp.K + col] : 0.0f;
  }
  const int nact = 2 * rp.rows * 32;
  for (int e = blk * kThreads + tid; e < nact; e += rp.n_blocks * kThreads) {
    const int net = e / (rp.rows * 32), rem = e - net * rp.rows * 32;
    const int j = rem / rp.rows, row = rem - j * rp.rows;   // consecutive threads -> consecutive rows of one j
    rp.w1at[net][rem] = j < rp.A ? rp.src[net][(size_t)row * rp.K + rp.O + j] : 0.0f;
  }
}
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_repack(RepackArgs rp) { repack_rows(rp, blockIdx.x, threadIdx.x); }
#endif

struct GatherArgs {
  const float* rb_obs; const float* rb_obs2; const float* rb_act; const float* rb_rew; const float* rb_done;
  const int* idx_table;   // [rows][B]
  int idx_rows;           // rows in the table (>=1)
  int use_dev;            // 1: iteration / table row from DevState (graph replay); 0: host values
  long long host_it; int host_row;
  float* X0; float* XP; float* X2; float* rew; float* done;
  int B, O, A, ldx;
  DevState* st;
  int bookkeeping;        // 1: block 0 performs prologue_duties (fused gather+step flows)
  int advance_counters;
  StepHyper hp;
  NoiseArgs nz;
  RepackArgs rp;
  int n_gather_blocks;
  int lookahead;          // use_dev: stage the minibatch of iteration it_next + lookahead (table row seq_next + lookahead)
};

// rows [4*blk, 4*blk+4) of the minibatch of iteration `it` (index-table row `trow`) + their noise
__device__ __forceinline__ void gather_block(const GatherArgs& a, int blk, long long it, int trow, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int r0 = blk * 4;
  const int r = r0 + wave;
  const bool row = r < a.B;
  // one trip of dwordx4 loads covers the row (O % 4 == 0, O <= 512): loads -> noise (VALU work under the HBM latency
  // of the random rows) -> stores. Other shapes: the row is copied before the noise.
  const bool one_trip = (a.O & 3) == 0 && a.O <= 512;
  const long long src = row ? a.idx_table[(size_t)trow * a.B + r] : 0;
  const float* __restrict__ so = a.rb_obs + (size_t)src * a.O;
  const float* __restrict__ so2 = a.rb_obs2 + (size_t)src * a.O;
  float* __restrict__ d0 = a.X0 + (size_t)(row ? r : 0) * a.ldx;
  float* __restrict__ dp = a.XP + (size_t)(row ? r : 0) * a.ldx;
  float* __restrict__ d2 = a.X2 + (size_t)(row ? r : 0) * a.ldx;
  const int ka = lane * 4, kb = ka + 256;
  f32x4 va = {0.f, 0.f, 0.f, 0.f}, wa = va, vb = va, wb = va;
  float av = 0.f, rw = 0.f, dn = 0.f;
  if (row) {
    av = a.rb_act[(size_t)src * a.A + (lane < a.A ? lane : 0)];
    rw = a.rb_rew[src]; dn = a.rb_done[src];
    if (one_trip) {
      if (ka < a.O) { va = *(const f32x4*)(so + ka); wa = *(const f32x4*)(so2 + ka); }
      if (kb < a.O) { vb = *(const f32x4*)(so + kb); wb = *(const f32x4*)(so2 + kb); }
    } else if ((a.O & 3) == 0) {
      for (int k0 = 0; k0 < a.O; k0 += 512) {
        const int k1 = k0 + lane * 4, k2 = k1 + 256;
        f32x4 xa = {0.f, 0.f, 0.f, 0.f}, ya = xa, xb = xa, yb = xa;
        if (k1 < a.O) { xa = *(const f32x4*)(so + k1); ya = *(const f32x4*)(so2 + k1); }
        if (k2 < a.O) { xb = *(const f32x4*)(so + k2); yb = *(const f32x4*)(so2 + k2); }
        if (k1 < a.O) { *(f32x4*)(d0 + k1) = xa; *(f32x4*)(dp + k1) = xa; *(f32x4*)(d2 + k1) = ya; }
        if (k2 < a.O) { *(f32x4*)(d0 + k2) = xb; *(f32x4*)(dp + k2) = xb; *(f32x4*)(d2 + k2) = yb; }
      }
    } else {
      for (int k0 = 0; k0 < a.O; k0 += 128) {
        const int k1 = k0 + lane, k2 = k1 + 64;
        float xa = 0.f, ya = 0.f, xb = 0.f, yb = 0.f;
        if (k1 < a.O) { xa = so[k1]; ya = so2[k1]; }
        if (k2 < a.O) { xb = so[k2]; yb = so2[k2]; }
        if (k1 < a.O) { d0[k1] = xa; dp[k1] = xa; d2[k1] = ya; }
        if (k2 < a.O) { d0[k2] = xb; dp[k2] = xb; d2[k2] = yb; }
      }
    }
  }
  if (a.nz.seed != 0) {
    const int r1 = r0 + 4 < a.B ? r0 + 4 : a.B;
    if (r0 < a.B) fill_noise_rows(a.nz, it, trow, r0, r1, a.A, tid, kThreads);
  }
  if (row) {
    if (one_trip) {
      if (ka < a.O) { *(f32x4*)(d0 + ka) = va; *(f32x4*)(dp + ka) = va; *(f32x4*)(d2 + ka) = wa; }
      if (kb < a.O) { *(f32x4*)(d0 + kb) = vb; *(f32x4*)(dp + kb) = vb; *(f32x4*)(d2 + kb) = wb; }
    }
    // action columns + zero padding up to ldx (A <= 32 < 64 lanes)
    for (int k = a.O + lane; k < a.ldx; k += 64) {
      const int j = k - a.O;
      d0[k] = j < a.A ? av : 0.0f;
      if (j >= a.A) { dp[k] = 0.0f; d2[k] = 0.0f; }
    }
    if (lane == 0) { a.rew[r] = rw; a.done[r] = dn; }
  }
}

__device__ __forceinline__ void gather_main(const GatherArgs& a, int blk, int tid) {
  const long long it = a.use_dev ? a.st->it_next + a.lookahead : a.host_it;
  const int trow = a.use_dev ? (int)((a.st->seq_next + a.lookahead) % a.idx_rows) : a.host_row;
  gather_block(a, blk, it, trow, tid);
  if (a.bookkeeping && blk == 0 && tid == 0) prologue_duties(a.st, it, a.advance_counters, a.hp);
}
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_gather(GatherArgs a) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x >= a.n_gather_blocks) {  // spare blocks: weight repack (independent of the gather)
    repack_rows(a.rp, (int)blockIdx.x - a.n_gather_blocks, tid);
    return;
  }
  gather_main(a, (int)blockIdx.x, tid);
}
#endif
// the pipelined graph opens with the minibatches of its first TWO updates (the riding gathers look two updates ahead):
// one launch, blocks [0, na) -> a, [na, na + nb) -> b, the rest -> a's repack blocks
struct Gather2Args { GatherArgs a, b; };
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_gather2(Gather2Args g) {
  const int tid = threadIdx.x, blk = (int)blockIdx.x, na = g.a.n_gather_blocks, nb = g.b.n_gather_blocks;
  if (blk < na) gather_main(g.a, blk, tid);
  else if (blk < na + nb) gather_main(g.b, blk - na, tid);
  else repack_rows(g.a.rp, blk - na - nb, tid);
}
#endif

// Riders of the loss launch in graph replays (k_loss has B/4 blocks: three quarters of the chip idle). Blocks
// [n_loss_blocks, +n_gather) stage the NEXT update's minibatch into the other batch set (iteration it_next + 1,
// table row seq_next + 1: the counters advance when this update closes; the pipelined graph stages the update
// AFTER the next, g.lookahead = 2); one more block does THIS update's
// bookkeeping (nothing before the first weight-gradient tile reads what prologue_duties writes).
struct RideArgs {
  GatherArgs g;          // destination pointers = the other set; g.st / g.hp also serve the bookkeeping
  int n_loss_blocks;     // blocks of the loss itself (always set)
  int n_gather;          // 0: no gather rides (last update of a graph, eager flows)
  int bookkeeping;       // 1: block n_loss_blocks + n_gather runs prologue_duties for this update
};
__device__ __forceinline__ bool loss_rider(const RideArgs& r) {
  const int b = (int)blockIdx.x - r.n_loss_blocks;
  if (b < 0) return false;
  if (b < r.n_gather) {
    const DevState* st = r.g.st;
    gather_block(r.g, b, st->it_next + r.g.lookahead, (int)((st->seq_next + r.g.lookahead) % r.g.idx_rows), threadIdx.x);
  } else if (r.bookkeeping && threadIdx.x == 0) {
    prologue_duties(r.g.st, r.g.st->it_next, 1, r.g.hp);
  }
  return true;
}

// stand-alone bookkeeping for the flows that do not gather (host-staged minibatch, apply-only)
struct PrologueArgs {
  DevState* st; int use_dev; long long host_it; int advance_counters; int fill_noise; StepHyper hp;
  NoiseArgs nz; int B, A; int table_rows;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_prologue(PrologueArgs a) {
  const long long it = a.use_dev ? a.st->it_next : a.host_it;
  if (a.nz.seed != 0 && a.fill_noise) fill_noise_rows(a.nz, it, a.use_dev && a.nz.table ? (int)(a.st->seq_next % a.table_rows) : 0, 0, a.B, a.A, threadIdx.x, kThreads);
  if (threadIdx.x == 0) prologue_duties(a.st, it, a.advance_counters, a.hp);
}
#endif

// device-side (stream-ordered, no host sync) reset of the replay counters before a group of graph-replayed updates:
// iteration of the group's first update, index / noise table row 0
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void k_set_counters(DevState* st, long long it_next, long long seq_next) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { st->it_next = it_next; st->seq_next = seq_next; }
}
#endif

// replay ring scatter (training/replay_buffer.py:58-83): n staged rows -> ring rows (ptr+i) % cap
struct ScatterArgs {
  const float* s_obs; const float* s_obs2; const float* s_act; const float* s_rew; const float* s_done; const float* s_logp;
  float* rb_obs; float* rb_obs2; float* rb_act; float* rb_rew; float* rb_done; float* rb_logp;
  long long ptr, cap; int n, O, A;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_ring_write(ScatterArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= a.n) return;
  const long long dst = (a.ptr + i) % a.cap;
  for (int k = lane; k < a.O; k += 64) {
    a.rb_obs[(size_t)dst * a.O + k] = a.s_obs[(size_t)i * a.O + k];
    a.rb_obs2[(size_t)dst * a.O + k] = a.s_obs2[(size_t)i * a.O + k];
  }
  for (int k = lane; k < a.A; k += 64) a.rb_act[(size_t)dst * a.A + k] = a.s_act[(size_t)i * a.A + k];
  if (lane == 0) {
    a.rb_rew[dst] = a.s_rew[i];
    a.rb_done[dst] = a.s_done[i];
    a.rb_logp[dst] = a.s_logp ? a.s_logp[i] : 0.0f;
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// k_tiles: C[m][n] (+epilogue) = sum_k P(m,k) * Q(n,k) on 32x32 tiles, BK = 64
//   operand storage: KC  element (row,k) at base[row*ld + k]   (k contiguous)
//                    MC  element (row,k) at base[k*ld + row]   (row contiguous)
//   forward  Z = X W^T      : P = X (KC),  Q = W (KC)          epilogue bias + GELU -> H, GELU'
//   backward dH = dZ W      : P = dZ (KC), Q = W (MC)          epilogue * GELU'(z_prev) -> dZ_prev
//   weights  dW = dZ^T X    : P = dZ (MC), Q = X (MC)          epilogue store (bias grads: Q = ones)
// 4 waves: wave (wr,wc) owns a 16x16 quadrant; each lane feeds one P and one Q element per MFMA.
// k-slot mapping: MFMA step s of a 16-wide k group contracts k = 4*(lane>>4) + s, identically for
// both operands, so a KC fragment is ONE ds_read_b128.
// ---------------------------------------------------------------------------------------------
constexpr int TM = 32, TN = 32, BK = 64;
constexpr int KC_LD = BK + 4;   // 68 floats: 16B-aligned rows, spreads ds_read_b128 over banks
constexpr int MC_LD = TM + 4;   // 36 floats
constexpr int TILE_LDS = BK * MC_LD;  // 2304 floats >= 32*KC_LD (2176)

enum : int { EPI_GELU = 0, EPI_MULG = 1, EPI_STORE = 2 };
// GemmProb::act of an EPI_MULG problem: aux holds post-ReLU activations a, the product is multiplied by (a > 0) -- the conv data
// gradient of a layer whose col2im is the identity (round 6: one output pixel, kernel = the whole input: type_2's last layer)
enum : int { MULG_RELU_MASK = 1 };


// Operand fetch: ONE unconditional dwordx4 per lane and slot (addresses clamped into the matrix,
// out-of-range elements zeroed by selects) so that every load of a tile -- and of ALL k-tiles, see
// run_tile -- is in flight at once; a divergent bounds branch around each load would serialise them
// (each dependent L2/MALL round trip is ~0.7 us, the whole tile's MFMA work ~0.2 us).
// The clamped vector may over-read <= 12 bytes past a row end; all operands sit inside larger
// allocations (parameter arenas / 256-byte spaced workspace buffers), never at an allocation end.
template <bool MC>
__device__ __forceinline__ f32x4 tile_load1(const float* __restrict__ base, int ld, int row0, int rows, int k0,
                                            int K, int tid, int j) {
  if (!MC) {
    const int row = row0 + (tid >> 4) + 16 * j;
    const int k = k0 + (tid & 15) * 4;
    const int rc = row < rows ? row : rows - 1;
    const int kc = k < K ? k : 0;
    f32x4 v = *(const f32x4u*)(base + (size_t)rc * ld + kc);
    const bool rv = row < rows;
    v.x = (rv && k < K) ? v.x : 0.f;
    v.y = (rv && k + 1 < K) ? v.y : 0.f;
    v.z = (rv && k + 2 < K) ? v.z : 0.f;
    v.w = (rv && k + 3 < K) ? v.w : 0.f;
    return v;
  } else {
    const int k = k0 + (tid >> 3) + 32 * j;
    const int row = row0 + (tid & 7) * 4;
    const int kc = k < K ? k : 0;
    const int rc = row < rows ? row : 0;
    f32x4 v = *(const f32x4u*)(base + (size_t)kc * ld + rc);
    const bool kv = k < K;
    v.x = (kv && row < rows) ? v.x : 0.f;
    v.y = (kv && row + 1 < rows) ? v.y : 0.f;
    v.z = (kv && row + 2 < rows) ? v.z : 0.f;
    v.w = (kv && row + 3 < rows) ? v.w : 0.f;
    return v;
  }
}

__device__ __forceinline__ f32x4 gload4(const float* p) {  // global (not flat) dwordx4, 4-byte aligned
  return *(const __attribute__((address_space(1))) f32x4u*)p;
}

// LDS image of a staged k-tile is ALWAYS [32 rows][KC_LD] with k contiguous, whatever the operand's
// layout in memory: an MC operand (rows contiguous in memory) is transposed on the way in (4 scalar
// ds_write_b32 per vector), so that every MFMA fragment is ONE ds_read_b128 instead of four ds_read_b32.
// The 16-byte quads of a row are stored SWIZZLED: quad q of row r sits at quad q ^ ((r >> 3) & 3). With the plain
// layout the transposing stores of a wave (rows 4j + e, j = 0..7, four consecutive k) fall into 16 of the 32 banks --
// a 4-way conflict on every one of the 16 stores per thread and step, and the LDS is ONE unit per CU shared by the four
// workgroups resident there (k_conv_dw: the staging, not the MFMAs, was the time; round 3). The XOR moves rows 8..15,
// 16..23, 24..31 to other quads, which spreads those stores over all banks and leaves the ds_read_b128 fragments (8
// consecutive rows per pass, (r >> 3) constant among them) and the 16-byte stores conflict-free as before.
__device__ __forceinline__ int lds_quad(int row, int quad) { return (quad ^ ((row >> 3) & 3)) << 2; }
template <bool MC>
__device__ __forceinline__ void tile_store_lds(float* lds, int tid, const f32x4& r0, const f32x4& r1) {
  if (!MC) {
    const int ra = tid >> 4, rb = ra + 16, q = tid & 15;
    *(f32x4*)(lds + ra * KC_LD + lds_quad(ra, q)) = r0;
    *(f32x4*)(lds + rb * KC_LD + lds_quad(rb, q)) = r1;
  } else {
    const int k = tid >> 3, row = (tid & 7) * 4;     // rows row .. row+3 share (row >> 3); k + 32 is quad + 8: same swizzle
    float* p = lds + row * KC_LD + lds_quad(row, k >> 2) + (k & 3);
    p[0] = r0.x; p[KC_LD] = r0.y; p[2 * KC_LD] = r0.z; p[3 * KC_LD] = r0.w;
    p[32] = r1.x; p[KC_LD + 32] = r1.y; p[2 * KC_LD + 32] = r1.z; p[3 * KC_LD + 32] = r1.w;
  }
}

__device__ __forceinline__ f32x4 frag_read(const float* lds, int row, int kk, int g) {
  // (kk * 4 + g) ^ s == kk * 4 + (g ^ s) for s < 4: the k-group stays an immediate offset of ONE per-thread base address
  return *(const f32x4*)(lds + row * KC_LD + ((g ^ ((row >> 3) & 3)) << 2) + kk * 16);
}

template <bool P_MC, bool Q_MC>
__device__ __forceinline__ void tile_mma(const float* ps, const float* qs, int prow, int qrow, int g, f32x4& acc0,
                                         f32x4& acc1) {
#pragma unroll
  for (int kk = 0; kk < BK / 16; ++kk) {
    const f32x4 p = frag_read(ps, prow, kk, g);
    const f32x4 q = frag_read(qs, qrow, kk, g);
    // D[row = n][col = m]: lane holds n = 4*g + reg, m = lane & 15
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.x, p.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.y, p.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.z, p.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q.w, p.w, acc1, 0, 0, 0);
  }
}

// one GEMM problem (all tiles of one matrix product)
struct GemmProb {
  const float* P; const float* Q;
  float* C0; float* C1;
  const float* aux;     // EPI_GELU: bias[N]; EPI_MULG: G[M x ldaux] (act == MULG_RELU_MASK: post-ReLU activations whose SIGN is the factor)
  int ldp, ldq, ldc, ldaux;
  int M, N, K;
  int tiles_n;          // n-tiles per m-tile row
  int tile_end;         // exclusive end of this problem's block range inside its stage
  const MirrorDesc* mir;  // weight-gradient tiles with the fused optimiser: packed copies of the tensor to refresh (or nullptr)
  int act;                // EPI_GELU stages: hidden activation of this problem's net (ACT_*, dsact_math.h)
  int mzero;              // EPI_STORE: rows m >= mzero > 0 are structurally zero (policy_std_type "parameter": the log-std half of
                          // the policy's output-layer weight) -- their gradient is stored as 0, so the optimiser leaves them at 0
};

// Optimiser fused into the weight-gradient tiles (single-GPU path): every parameter element is the
// output of exactly ONE dW/db tile (K = batch is reduced inside the tile), so the tile applies Adam
// (and Polyak on delayed steps) to its own 32x32 block right after writing the gradient and the
// separate streaming pass over the arenas (k_adam) disappears. Arena element index of an output =
// (C0 - grads) + m*ldc + n because grads / online / adam_m / adam_v / target mirror each other.
struct FusedOpt {
  DevState* st;            // nullptr: plain store (data-parallel path keeps k_adam after the all-reduce)
  float* online; float* target; float* adam_m; float* adam_v; float* grads;
  long long n_q2, n_online3, n_total;
  float b1w, beta2, b2w, eps, polyak, one_minus_polyak;
  int auto_alpha;
  // Graph replays whose gather rides in the previous update's loss launch (no per-step k_gather, hence no per-step
  // repack pass): the tiles that update the Q nets' first-layer weights also refresh the zero-padded copies the next
  // forward reads (mir_w / mir_wt) and the transposed action columns the NEXT update's k_heads_bwd reads (mir_at: the
  // other batch set's copy -- this update's own copy is still being read while these tiles run). mir_n == 0: off.
  int mir_n;                          // number of Q nets (0 = off)
  int mir_ldp, mir_O, mir_A, mir_rows;
  long long mir_lo[2];                // arena index of W0 of q1 / q2
  float* mir_w[2]; float* mir_wt[2];  // padded copies [rows x ldp]: online, target
  float* mir_at[2];                   // [32][rows]
};

constexpr int kMaxPrefetchTiles = 7;  // K <= 448 is fetched completely up front (112 VGPRs)
// dynamic LDS a tile kernel needs for contraction length K
inline size_t tile_lds_bytes(int K) {
  const int T = (K + BK - 1) / BK;
  return (size_t)(T <= kMaxPrefetchTiles ? T : 2) * 2 * TILE_LDS * sizeof(float);
}

// per-thread operand cursor: the two (clamped) element pointers of k-tile 0 and what the fast path
// needs to know. Full tiles (uniform test) are plain dwordx4 loads at pointer + tile offset.
template <bool MC>
struct OpCursor {
  const float* p0; const float* p1;  // slot 0 / slot 1 of k-tile 0
  size_t step;                       // floats between consecutive k-tiles
  bool v0, v1;                       // slot row valid (KC) -- always true when the tile is row-full
};

template <bool MC>
__device__ __forceinline__ OpCursor<MC> make_cursor(const float* base, int ld, int row0, int rows, int tid) {
  OpCursor<MC> c;
  if (!MC) {
    const int r_0 = row0 + (tid >> 4), r_1 = r_0 + 16;
    c.v0 = r_0 < rows; c.v1 = r_1 < rows;
    c.p0 = base + (size_t)(c.v0 ? r_0 : rows - 1) * ld + (tid & 15) * 4;
    c.p1 = base + (size_t)(c.v1 ? r_1 : rows - 1) * ld + (tid & 15) * 4;
    c.step = BK;
  } else {
    const int col = row0 + (tid & 7) * 4;
    c.v0 = c.v1 = true;
    c.p0 = base + (size_t)(tid >> 3) * ld + col;
    c.p1 = c.p0 + (size_t)32 * ld;
    c.step = (size_t)BK * ld;
  }
  return c;
}

// TS > 0: compile-time specialisation for "clean" problems (M, N multiples of 32, K == TS*64): the load
// block is straight-line code -- 4*TS dwordx4 loads back to back, no bounds logic, no branches.
template <bool P_MC, bool Q_MC, int EPI, int TS = 0>
__device__ __forceinline__ void run_tile(const GemmProb& t, int m0, int n0, float* lds, long long* tl_buf = nullptr,
                                         int tl_slot = 0, const FusedOpt* fo = nullptr) {
  TL_DECL
  TL_STAMP();  // 0: tile start (problem decoded)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, g = lane >> 4;
  // LDS image: k-tile t occupies [t*2*TILE_LDS, (t+1)*2*TILE_LDS): P tile then Q tile. When the whole
  // K range is prefetched (T <= kMaxPrefetchTiles) every tile has its own slot and ONE barrier
  // separates staging from the MFMA run; otherwise two slots are used as a double buffer.
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int T = (t.K + BK - 1) / BK;
  const int Tfull = t.K / BK;                      // k-tiles that need no k mask
  // a row-full tile needs no row mask; MC operands additionally need it for the vector not to straddle
  const bool p_rows_full = m0 + TM <= t.M, q_rows_full = n0 + TN <= t.N;
  const OpCursor<P_MC> pc = make_cursor<P_MC>(t.P, t.ldp, m0, t.M, tid);
  const OpCursor<Q_MC> qc = make_cursor<Q_MC>(t.Q, t.ldq, n0, t.N, tid);
  const bool p_fast = P_MC ? p_rows_full : true, q_fast = Q_MC ? q_rows_full : true;
  // output coordinates + epilogue operands (bias / GELU') are fetched now, not after the MFMA loop
  const int m = m0 + wr * 16 + i;
  const int n = n0 + wc * 16 + 4 * g;
  const bool in_range = m < t.M && n < t.N;
  const bool full = n + 3 < t.N;
  f32x4 epv = {0.f, 0.f, 0.f, 0.f};
  if (EPI == EPI_GELU) { if (in_range && full) epv = *(const f32x4u*)(t.aux + n); }
  else if (EPI == EPI_MULG) { if (in_range && full) epv = *(const f32x4u*)(t.aux + (size_t)m * t.ldaux + n); }
  // fused optimiser: this lane's 4 parameters and their moments, fetched now
  const bool fused = EPI == EPI_STORE && fo != nullptr && fo->st != nullptr;
  long long oi = 0;
  bool o_delayed = false, o_upd = false, o_tvec = false;
  float o_ss = 0.f, o_bc2 = 1.f;
  f32x4 op = {0.f, 0.f, 0.f, 0.f}, om = op, ov = op, ot = op;
  if (fused && in_range) {
    oi = (long long)((t.C0 + (size_t)m * t.ldc + n) - fo->grads);
    o_delayed = fo->st->do_delayed != 0;
    const bool is_q = oi < fo->n_q2;  // a tile never straddles nets: one tensor per problem
    o_upd = is_q || o_delayed;
    o_ss = is_q ? fo->st->ss_q : fo->st->ss_pi;
    o_bc2 = is_q ? fo->st->bc2_q : fo->st->bc2_pi;
    if (o_upd && full) {
      op = *(const f32x4u*)(fo->online + oi); om = *(const f32x4u*)(fo->adam_m + oi); ov = *(const f32x4u*)(fo->adam_v + oi);
      o_tvec = o_delayed;
      if (o_tvec) ot = *(const f32x4u*)(fo->target + oi);
    }
  }
#define DSACT_LOAD_TILE(IT, P0, P1, Q0, Q1)                                                         \
  do {                                                                                              \
    if ((IT) < Tfull && p_fast) {                                                                   \
      P0 = *(const f32x4u*)(pc.p0 + (size_t)(IT) * pc.step);                                        \
      P1 = *(const f32x4u*)(pc.p1 + (size_t)(IT) * pc.step);                                        \
      if (!P_MC && !p_rows_full) { if (!pc.v0) P0 = f32x4{0.f, 0.f, 0.f, 0.f}; if (!pc.v1) P1 = f32x4{0.f, 0.f, 0.f, 0.f}; } \
    } else {                                                                                        \
      P0 = tile_load1<P_MC>(t.P, t.ldp, m0, t.M, (IT) * BK, t.K, tid, 0);                           \
      P1 = tile_load1<P_MC>(t.P, t.ldp, m0, t.M, (IT) * BK, t.K, tid, 1);                           \
    }                                                                                               \
    if ((IT) < Tfull && q_fast) {                                                                   \
      Q0 = *(const f32x4u*)(qc.p0 + (size_t)(IT) * qc.step);                                        \
      Q1 = *(const f32x4u*)(qc.p1 + (size_t)(IT) * qc.step);                                        \
      if (!Q_MC && !q_rows_full) { if (!qc.v0) Q0 = f32x4{0.f, 0.f, 0.f, 0.f}; if (!qc.v1) Q1 = f32x4{0.f, 0.f, 0.f, 0.f}; } \
    } else {                                                                                        \
      Q0 = tile_load1<Q_MC>(t.Q, t.ldq, n0, t.N, (IT) * BK, t.K, tid, 0);                           \
      Q1 = tile_load1<Q_MC>(t.Q, t.ldq, n0, t.N, (IT) * BK, t.K, tid, 1);                           \
    }                                                                                               \
  } while (0)
  if constexpr (TS > 0) {
    f32x4 pr[TS][2], qr[TS][2];
#pragma unroll
    for (int it = 0; it < TS; ++it) {
      pr[it][0] = gload4(pc.p0 + (size_t)it * pc.step);
      pr[it][1] = gload4(pc.p1 + (size_t)it * pc.step);
      qr[it][0] = gload4(qc.p0 + (size_t)it * qc.step);
      qr[it][1] = gload4(qc.p1 + (size_t)it * qc.step);
    }
    TL_STAMP();  // 1: all loads issued
#pragma unroll
    for (int it = 0; it < TS; ++it) {
      tile_store_lds<P_MC>(lds + it * 2 * TILE_LDS, tid, pr[it][0], pr[it][1]);
      tile_store_lds<Q_MC>(lds + it * 2 * TILE_LDS + TILE_LDS, tid, qr[it][0], qr[it][1]);
    }
    __syncthreads();
    TL_STAMP();  // 2: every k-tile landed and staged
#pragma unroll
    for (int it = 0; it < TS; ++it)
      tile_mma<P_MC, Q_MC>(lds + it * 2 * TILE_LDS, lds + it * 2 * TILE_LDS + TILE_LDS, wr * 16 + i, wc * 16 + i, g, acc0, acc1);
  } else if (T <= kMaxPrefetchTiles) {
    f32x4 pr[kMaxPrefetchTiles][2], qr[kMaxPrefetchTiles][2];
#pragma unroll
    for (int it = 0; it < kMaxPrefetchTiles; ++it) {
      if (it < T) DSACT_LOAD_TILE(it, pr[it][0], pr[it][1], qr[it][0], qr[it][1]);
    }
    TL_STAMP();  // 1: all loads issued
#pragma unroll
    for (int it = 0; it < kMaxPrefetchTiles; ++it) {
      if (it < T) {
        tile_store_lds<P_MC>(lds + it * 2 * TILE_LDS, tid, pr[it][0], pr[it][1]);
        tile_store_lds<Q_MC>(lds + it * 2 * TILE_LDS + TILE_LDS, tid, qr[it][0], qr[it][1]);
      }
    }
    __syncthreads();
    TL_STAMP();  // 2: every k-tile landed and staged
#pragma unroll
    for (int it = 0; it < kMaxPrefetchTiles; ++it) {
      if (it < T)
        tile_mma<P_MC, Q_MC>(lds + it * 2 * TILE_LDS, lds + it * 2 * TILE_LDS + TILE_LDS, wr * 16 + i, wc * 16 + i, g, acc0, acc1);
    }
  } else {
    float* Ps[2] = {lds, lds + 2 * TILE_LDS};
    float* Qs[2] = {lds + TILE_LDS, lds + 3 * TILE_LDS};
    f32x4 p0, p1, q0, q1;
    DSACT_LOAD_TILE(0, p0, p1, q0, q1);
    for (int it = 0; it < T; ++it) {
      const int b = it & 1;
      tile_store_lds<P_MC>(Ps[b], tid, p0, p1);
      tile_store_lds<Q_MC>(Qs[b], tid, q0, q1);
      __syncthreads();
      if (it + 1 < T) DSACT_LOAD_TILE(it + 1, p0, p1, q0, q1);
      tile_mma<P_MC, Q_MC>(Ps[b], Qs[b], wr * 16 + i, wc * 16 + i, g, acc0, acc1);
    }
  }
#undef DSACT_LOAD_TILE
  f32x4 acc = acc0 + acc1;
  TL_STAMP();  // 3: MFMA loop done
  if (!in_range) { TL_FLUSH(tl_buf, tl_slot); return; }
  if (EPI == EPI_GELU) {
    f32x4 h, gd, zv;
#pragma unroll
    for (int e = 0; e < 4; ++e) zv[e] = acc[e] + (full ? epv[e] : (n + e < t.N ? t.aux[n + e] : 0.0f));
    act4(t.act, zv, h, gd);
    float* c0 = t.C0 + (size_t)m * t.ldc + n;
    float* c1 = t.C1 + (size_t)m * t.ldc + n;
    if (full) { *(f32x4u*)c0 = h; *(f32x4u*)c1 = gd; }
    else for (int e = 0; e < 4 && n + e < t.N; ++e) { c0[e] = h[e]; c1[e] = gd[e]; }
  } else if (EPI == EPI_MULG) {
    float* c0 = t.C0 + (size_t)m * t.ldc + n;
    const bool mask = t.act == MULG_RELU_MASK;   // (a select, not a product: a masked element is +0 like col2im's / k_feat_bwd's)
    if (full) {
      f32x4 o = acc * epv;
      if (mask) for (int e = 0; e < 4; ++e) o[e] = epv[e] > 0.f ? acc[e] : 0.f;
      *(f32x4u*)c0 = o;
    } else {
      const float* gp = t.aux + (size_t)m * t.ldaux + n;
      for (int e = 0; e < 4 && n + e < t.N; ++e) c0[e] = mask ? (gp[e] > 0.f ? acc[e] : 0.f) : acc[e] * gp[e];
    }
  } else {
    float* c0 = t.C0 + (size_t)m * t.ldc + n;
    if (t.mzero > 0 && m >= t.mzero) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (full) *(f32x4u*)c0 = acc;
    else for (int e = 0; e < 4 && n + e < t.N; ++e) c0[e] = acc[e];
    if (fused && o_upd) {
      if (full) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = op[e], me = om[e], ve = ov[e];
          adam_update(pe, me, ve, acc[e], fo->b1w, fo->beta2, fo->b2w, o_ss, o_bc2, fo->eps);
          op[e] = pe; om[e] = me; ov[e] = ve;
          if (o_tvec) ot[e] = polyak_update(ot[e], pe, fo->polyak, fo->one_minus_polyak);
        }
        *(f32x4u*)(fo->online + oi) = op; *(f32x4u*)(fo->adam_m + oi) = om; *(f32x4u*)(fo->adam_v + oi) = ov;
        if (o_tvec) *(f32x4u*)(fo->target + oi) = ot;
      } else {
        for (int e = 0; e < 4 && n + e < t.N; ++e) {
          float pe = fo->online[oi + e], me = fo->adam_m[oi + e], ve = fo->adam_v[oi + e];
          adam_update(pe, me, ve, acc[e], fo->b1w, fo->beta2, fo->b2w, o_ss, o_bc2, fo->eps);
          fo->online[oi + e] = pe; fo->adam_m[oi + e] = me; fo->adam_v[oi + e] = ve;
          op[e] = pe;
          if (o_delayed) { ot[e] = polyak_update(fo->target[oi + e], pe, fo->polyak, fo->one_minus_polyak); fo->target[oi + e] = ot[e]; }
        }
      }
      // fragment-major copies the chain kernels read (dsact_chain.h)
      if (t.mir) mirror_store4(*t.mir, m, n, t.N, op, o_delayed, ot);
      // first-layer weights of a Q net: refresh the padded / transposed copies (see FusedOpt::mir_*)
      for (int q = 0; q < fo->mir_n; ++q) {
        if (t.C0 != fo->grads + fo->mir_lo[q]) continue;   // uniform: one tensor per problem
        float* mw = fo->mir_w[q] + (size_t)m * fo->mir_ldp + n;
        float* mt = fo->mir_wt[q] + (size_t)m * fo->mir_ldp + n;
        if (full) {
          *(f32x4*)mw = op;
          if (o_delayed) *(f32x4*)mt = ot;
        } else {
          for (int e = 0; e < 4 && n + e < t.N; ++e) { mw[e] = op[e]; if (o_delayed) mt[e] = ot[e]; }
        }
        for (int e = 0; e < 4; ++e) {
          const int j = n + e - fo->mir_O;
          if (j >= 0 && j < fo->mir_A && n + e < t.N) fo->mir_at[q][(size_t)j * fo->mir_rows + m] = op[e];
        }
      }
    }
  }
  TL_STAMP();  // 4: epilogue stores issued
  TL_FLUSH(tl_buf, tl_slot);
}

// XCD-aware block -> logical tile map. The dispatcher places block b on XCD b % 8 and each XCD has a
// private, non-coherent L2. Tiles are listed (problem, m-tile, n-tile), so giving every XCD one
// CONTIGUOUS chunk of that list confines a problem's weight / activation slabs to 1-2 XCDs instead
// of having all 8 L2s fetch all of them through the fabric. Bijective for any grid size; placement
// only affects speed, never results.
__device__ __forceinline__ int xcd_logical_block(int b, int nb) {
  const int xcd = b & 7, slot = b >> 3;
  const int q = nb >> 3, r = nb & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ---- stage launch form 1: <= kMaxProb problems described in the kernel arguments ----------------
constexpr int kMaxProb = 8;   // 4 chains x 2 trunks (CNN nets) per stage
struct StageArgs {
  GemmProb p[kMaxProb];
  int n_prob;
  int n_stage_blocks;      // blocks [0, n_stage_blocks) run the problems above ...
  const GemmProb* extra;   // ... the remaining blocks run per-tile table entries (weight gradients that
  int n_extra;             //     are independent of this stage and would otherwise idle-wait for it)
  FusedOpt fo;             // optimiser applied by those tiles (fo.st == nullptr: plain gradient store)
  long long* timeline;     // DSACT_TIMELINE builds only (else unused)
};

template <bool P_MC, bool Q_MC, int EPI, int TS = 0>
__global__ void __launch_bounds__(kThreads) k_stage(StageArgs s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // tile_lds_bytes(max K of the launch)
  const int b = xcd_logical_block(blockIdx.x, gridDim.x);
  if (b >= s.n_stage_blocks) {  // ride-along weight-gradient tile (MC x MC, plain store)
    const GemmProb g = s.extra[b - s.n_stage_blocks];
    run_tile<true, true, EPI_STORE>(g, g.tiles_n, g.tile_end, lds, nullptr, 0, &s.fo);
    return;
  }
  int pi = 0;
#pragma unroll
  for (int q = 0; q + 1 < kMaxProb; ++q)
    if (q + 1 < s.n_prob && b >= s.p[q].tile_end) pi = q + 1;
  const GemmProb& g = s.p[pi];
  const int local = b - (pi ? s.p[pi - 1].tile_end : 0);
  const int mt = local / g.tiles_n, nt = local - mt * g.tiles_n;
  run_tile<P_MC, Q_MC, EPI, TS>(g, mt * TM, nt * TN, lds, s.timeline, (int)blockIdx.x);
}

// ---- large batches: 64x64 output tiles, 512 threads ---------------------------------------------------
// At batch >= 512 a stage has >= 4x the tiles of the batch-256 case and is bound by operand traffic (a 32x32 tile
// moves 64 KB per 0.5 MFLOP). A 64x64 tile halves the bytes per FLOP. 8 waves: wave (wr = w>>2, wc = w&3) owns
// rows wr*32..+31 x columns wc*16..+15 (two accumulator tiles sharing one weight fragment). k-tiles of 64 are
// double-buffered through LDS with the next tile's global loads in flight; the masks of the (padded) k tail are
// applied at LDS-store time. Requirements (checked by the host): M % 64 == 0, N % 64 == 0, K % 4 == 0.
constexpr int kThreads64 = 512;
constexpr int TILE64_LDS = 64 * KC_LD;   // floats per staged operand tile
inline size_t tile64_lds_bytes() { return (size_t)2 * 2 * TILE64_LDS * sizeof(float); }

// raw loads of one staged operand tile; k quads / k rows beyond K are clamped to k = 0 (masked at store time)
template <bool MC>
__device__ __forceinline__ void tile64_load(const float* __restrict__ base, int ld, int row0, int kt, int K, int tid, f32x4& r0, f32x4& r1) {
  if (!MC) {        // (row, k) at base[row*ld + k]: rows (tid>>4) and +32, k quad tid&15
    const int k = kt * BK + (tid & 15) * 4;
    const float* p = base + (size_t)(row0 + (tid >> 4)) * ld + (k < K ? k : 0);
    r0 = gload4(p);
    r1 = gload4(p + (size_t)32 * ld);
  } else {          // (row, k) at base[k*ld + row]: row quad tid&15, k = (tid>>4) and +32
    const int ka = kt * BK + (tid >> 4), kb = ka + 32;
    const float* p = base + row0 + (tid & 15) * 4;
    r0 = gload4(p + (size_t)(ka < K ? ka : 0) * ld);
    r1 = gload4(p + (size_t)(kb < K ? kb : 0) * ld);
  }
}
template <bool MC>
__device__ __forceinline__ void tile64_store_lds(float* lds, int tid, const f32x4& r0, const f32x4& r1) {
  if (!MC) {
    const int ra = tid >> 4, rb = ra + 32, q = tid & 15;
    *(f32x4*)(lds + ra * KC_LD + lds_quad(ra, q)) = r0;
    *(f32x4*)(lds + rb * KC_LD + lds_quad(rb, q)) = r1;
  } else {
    const int k = tid >> 4, row = (tid & 15) * 4;
    float* p = lds + row * KC_LD + lds_quad(row, k >> 2) + (k & 3);
    p[0] = r0.x; p[KC_LD] = r0.y; p[2 * KC_LD] = r0.z; p[3 * KC_LD] = r0.w;
    p[32] = r1.x; p[KC_LD + 32] = r1.y; p[2 * KC_LD + 32] = r1.z; p[3 * KC_LD + 32] = r1.w;
  }
}

template <bool Q_MC, int EPI>
__device__ __forceinline__ void run_tile64(const GemmProb& t, int m0, int n0, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int i = lane & 15, g = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int T = (t.K + BK - 1) / BK;
  // k validity of this thread's quads: KC operand: quad (tid&15) of the tile; MC operand: rows k = tid>>4, +32
  auto kvalid_kc = [&](int kt) { return kt * BK + (tid & 15) * 4 < t.K; };
  auto kvalid_mc = [&](int kt, int j) { return kt * BK + (tid >> 4) + 32 * j < t.K; };
  auto load = [&](int kt, f32x4& p0, f32x4& p1, f32x4& q0, f32x4& q1) {
    tile64_load<false>(t.P, t.ldp, m0, kt, t.K, tid, p0, p1);
    tile64_load<Q_MC>(t.Q, t.ldq, n0, kt, t.K, tid, q0, q1);
  };
  // epilogue operands first (bias / GELU' quads of this lane's two output blocks)
  const int n = n0 + wc * 16 + 4 * g;
  f32x4 epv[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = m0 + wr * 32 + mb * 16 + i;
    if (EPI == EPI_GELU) epv[mb] = gload4(t.aux + n);
    else if (EPI == EPI_MULG) epv[mb] = gload4(t.aux + (size_t)m * t.ldaux + n);
    else epv[mb] = zero;   // EPI_STORE: plain store (round 4: the conv data gradient's dCol = dY . W on 64 x 64 tiles)
  }
  f32x4 acc[2] = {zero, zero};
  f32x4 p0, p1, q0, q1;
  load(0, p0, p1, q0, q1);
  for (int kt = 0; kt < T; ++kt) {
    float* Ps = lds + (kt & 1) * 2 * TILE64_LDS;
    float* Qs = Ps + TILE64_LDS;
    if (!kvalid_kc(kt)) { p0 = zero; p1 = zero; if (!Q_MC) { q0 = zero; q1 = zero; } }
    if (Q_MC) { if (!kvalid_mc(kt, 0)) q0 = zero; if (!kvalid_mc(kt, 1)) q1 = zero; }
    tile64_store_lds<false>(Ps, tid, p0, p1);
    tile64_store_lds<Q_MC>(Qs, tid, q0, q1);
    __syncthreads();
    if (kt + 1 < T) load(kt + 1, p0, p1, q0, q1);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const f32x4 q = frag_read(Qs, wc * 16 + i, kk, g);
      const f32x4 pa = frag_read(Ps, wr * 32 + i, kk, g);
      const f32x4 pb = frag_read(Ps, wr * 32 + 16 + i, kk, g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], pa[e], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], pb[e], acc[1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = m0 + wr * 32 + mb * 16 + i;
    float* c0 = t.C0 + (size_t)m * t.ldc + n;
    if (EPI == EPI_GELU) {
      f32x4 hv, gd;
      act4(t.act, acc[mb] + epv[mb], hv, gd);
      *(f32x4u*)c0 = hv;
      *(f32x4u*)(t.C1 + (size_t)m * t.ldc + n) = gd;
    } else if (EPI == EPI_MULG) {
      f32x4 o = acc[mb] * epv[mb];
      if (t.act == MULG_RELU_MASK) for (int e = 0; e < 4; ++e) o[e] = epv[mb][e] > 0.f ? acc[mb][e] : 0.f;   // (a select: +0 where masked)
      *(f32x4u*)c0 = o;
    } else {
      *(f32x4u*)c0 = acc[mb];
    }
  }
}

template <bool Q_MC, int EPI>
__global__ void __launch_bounds__(kThreads64) k_stage64(StageArgs s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = xcd_logical_block(blockIdx.x, gridDim.x);
  int pi = 0;
#pragma unroll
  for (int q = 0; q + 1 < kMaxProb; ++q)
    if (q + 1 < s.n_prob && b >= s.p[q].tile_end) pi = q + 1;
  const GemmProb& g = s.p[pi];
  const int local = b - (pi ? s.p[pi - 1].tile_end : 0);
  const int mt = local / g.tiles_n, nt = local - mt * g.tiles_n;
  run_tile64<Q_MC, EPI>(g, mt * 64, nt * 64, lds);
}

// ---- stage launch form 2: many small problems (weight / bias gradients) from a device table ------
// one GemmProb PER TILE (tiles_n / tile_end are reused as the tile origin m0 / n0): a single
// dependent load per block. Every problem is an MC x MC product with a plain store.
struct TableArgs {
  const GemmProb* tiles; int n_tiles;
  FusedOpt fo;
  int finalize;   // 1: one extra block closes the update (alpha step, mean_std commit, counters)
  long long* timeline;   // DSACT_TIMELINE builds only
};

// end-of-update duties of the fused path: Adam on log_alpha, commit of the mean_std EMA and of the
// iteration / sequence counters (k_adam does the same on the unfused path)
__device__ __forceinline__ void finalize_update_s(DevState* stp, float* online, float* adam_m, float* adam_v, const float* grads,
                                                  long long n_total, int auto_alpha, float b1w, float beta2, float b2w, float eps) {
  const int do_delayed = stp->do_delayed;
  const float ss_alpha = stp->ss_alpha, bc2_alpha = stp->bc2_alpha;
  const long long it_cur = stp->it_cur, seq = stp->seq_next;
  const long long i = n_total - 1;
  if (do_delayed && auto_alpha) {
    float p = online[i], m = adam_m[i], v = adam_v[i];
    adam_update(p, m, v, grads[i], b1w, beta2, b2w, ss_alpha, bc2_alpha, eps);
    online[i] = p; adam_m[i] = m; adam_v[i] = v;
  }
  stp->ms1 = grads[n_total]; stp->ms2 = grads[n_total + 1]; stp->ms_init = 1;
  stp->it_next = it_cur + 1;
  stp->seq_next = seq + 1;
  stp->tag_seq += 1;
}
__device__ __forceinline__ void finalize_update(const FusedOpt& fo) {
  finalize_update_s(fo.st, fo.online, fo.adam_m, fo.adam_v, fo.grads, fo.n_total, fo.auto_alpha, fo.b1w, fo.beta2, fo.b2w, fo.eps);
}

// workgroup barrier that orders LDS only. (__syncthreads() may ALSO drain vmcnt -- every global load and store in
// flight: prefetched operands, an epilogue's stores -- when the compiler sees memory operations around it, and may
// emit a bare s_barrier when it does not: code that needs either behaviour says so explicitly, here and in
// stores_acked_barrier() of dsact_chain.h.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_stage_table(TableArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_tiles) {
    if (a.finalize && threadIdx.x == 0) finalize_update(a.fo);
    return;
  }
  const GemmProb g = a.tiles[xcd_logical_block(blockIdx.x, a.n_tiles)];
  run_tile<true, true, EPI_STORE>(g, g.tiles_n, g.tile_end, lds, a.timeline, (int)blockIdx.x, &a.fo);
}
#endif

// ---------------------------------------------------------------------------------------------
// row-vector helpers for the narrow layers (N_out = 2 or 2A, K = 2A): one wave per row, the row lives
// in registers (float4 per lane per 256-chunk; NCH = ceil(W/256) chunks, template parameter so that
// register arrays are statically indexed). Weight rows are fetched in GROUPS of independent
// dwordx4 loads -- a dependent L2/MALL round trip costs ~0.7 us, so N_out sequential dot products
// would cost N_out round trips; a group costs one.
// ---------------------------------------------------------------------------------------------
// RAW row fetch: the elements beyond W are NOT zeroed here -- a select on a just-loaded register makes
// the wave wait for that load before it can issue the next one. Callers apply row_mask() after all
// their loads are in flight.
template <int NCH>
__device__ __forceinline__ void row_load(const float* x, int W, int lane, f32x4 (&r)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 256 + lane * 4;
    r[c] = *(const f32x4u*)(x + (k < W ? k : 0));   // may over-read <= 12 B inside the workspace
  }
}
template <int NCH>
__device__ __forceinline__ void row_mask(f32x4 (&r)[NCH], int W, int lane) {
  if ((W & 255) == 0 && NCH * 256 == ((W + 255) & ~255)) return;  // every lane of every chunk is inside the row
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 256 + lane * 4;
    r[c].x = k < W ? r[c].x : 0.f;
    r[c].y = k + 1 < W ? r[c].y : 0.f;
    r[c].z = k + 2 < W ? r[c].z : 0.f;
    r[c].w = k + 3 < W ? r[c].w : 0.f;
  }
}

// out[q] = <h, w[(n0+q)*W .. ]> for q < G (rows clamped to n_out-1; the caller ignores the extras).
// h is zero beyond W, so clamped / over-read weight elements (finite parameters) contribute 0.
template <int NCH, int G>
__device__ __forceinline__ void row_dots(const f32x4 (&h)[NCH], const float* w, int W, int n0, int n_out,
                                         int lane, float (&out)[G]) {
  f32x4 wv[G][NCH];
#pragma unroll
  for (int q = 0; q < G; ++q) {
    const int n = n0 + q < n_out ? n0 + q : n_out - 1;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = c * 256 + lane * 4;
      wv[q][c] = *(const f32x4u*)(w + (size_t)n * W + (k < W ? k : 0));
    }
  }
  f32x4 hm[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) hm[c] = h[c];
  row_mask<NCH>(hm, W, lane);   // h comes from row_load (raw): zero what lies beyond the row, now that the loads are out
#pragma unroll
  for (int q = 0; q < G; ++q) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      s += hm[c].x * wv[q][c].x; s += hm[c].y * wv[q][c].y; s += hm[c].z * wv[q][c].z; s += hm[c].w * wv[q][c].w;
    }
    out[q] = wave_sum(s);
  }
}

// ---------------------------------------------------------------------------------------------
// k_heads: output layers of policy(obs), policy_target(obs2), q1(obs,act), q2(obs,act)
//   policy chains: logits -> (mu, raw) -> rsample (act_distribution_cls.py:44-54) -> action into
//   the action columns of XP (policy) / X2 (policy_target), log-prob.
//   grid (ceil(B/4), 4): y = 0 policy, 1 policy_target, 2 q1, 3 q2.
// ---------------------------------------------------------------------------------------------
struct HeadsArgs {
  const float* H[4];      // last hidden activations [B x W]
  const float* Wout[4];   // [n_out x W]
  const float* bout[4];
  int W, B, O, A, ldx;     // W: the widest last hidden layer (template dispatch); Wch: per chain
  int Wch[4];
  const float* eps_new; const float* eps_2;
  float* XP; float* X2;
  float* XPb; float* X2b;   // second destination of new_act / act2 (CNN nets: one input-row buffer per chain), or NULL
  float* logits_pi;   // [B x 2A] (mu | raw log-std) of policy(obs), kept for the backward
  float* logits_pit;  // [B x 2A] same for policy_target(obs2) (debug / parity only)
  float* logp_new; float* logp2;
  float* qout[2];     // raw (mean, pre-softplus std) of q1/q2(obs,act)  [B x 2]
  float* qstd[2];     // (softplus(raw std), d softplus / d raw) of the same          [B x 2]
  float* part_heads;  // [gridDim.x][2]: sum tanh(mu), sum sigma  (policy chain)
  const float* act_scale; const float* act_center;  // (hi-lo)/2, (hi+lo)/2
  float lo_ls, hi_ls;
  long long* timeline;
  int v1_stats;   // DSAC_V1 reports tanh(logits[...,0]) and logits[...,1] only (dsac_v1.py:145-146)
  int q_out_act, pi_out_act;   // output activations (dsact_math.h out_act_fwd): the stored outputs are POST-activation
  int pi_out_n;                // policy outputs the activation applies to: 2A, or A with policy_std_type "parameter" (log_std is a
                               // plain parameter there, networks/mlp.py:92-97)
  float* qdmean[2];            // OUT_ACT_GELU: d mean_y / d z of q1 / q2(obs, act) per row [B] (k_loss reads it), else nullptr
  float* pi_dact;              // OUT_ACT_GELU: d logit_y / d z of policy(obs) [B x 2A] (k_heads_bwd reads it), else nullptr
};

template <int NCH>
__global__ void __launch_bounds__(kThreads) k_heads(HeadsArgs a) {
  __shared__ float red[8];
  constexpr int G = NCH == 1 ? 12 : (NCH == 2 ? 8 : 4);
  TL_DECL
  TL_STAMP();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chain = blockIdx.y;
  const int r = blockIdx.x * 4 + wave;
  const bool active = r < a.B;
  float s_tanh = 0.f, s_sig = 0.f;
  if (active) {
    f32x4 h[NCH];
    const int W = a.Wch[chain];
    row_load<NCH>(a.H[chain] + (size_t)r * W, W, lane, h);
    if (chain >= 2) {
      float o[2];
      row_dots<NCH, 2>(h, a.Wout[chain], W, 0, 2, lane, o);
      if (lane == 0) {
        const float zm = o[0] + a.bout[chain][0], zr = o[1] + a.bout[chain][1];
        const float mean = out_act_fwd(a.q_out_act, zm), raw = out_act_fwd(a.q_out_act, zr);
        a.qout[chain - 2][2 * r] = mean;
        a.qout[chain - 2][2 * r + 1] = raw;
        a.qstd[chain - 2][2 * r] = softplus(raw);
        a.qstd[chain - 2][2 * r + 1] = softplus_grad(raw) * out_act_grad(a.q_out_act, raw, zr);   // d std / d (pre-activation output)
        if (a.qdmean[chain - 2]) a.qdmean[chain - 2][r] = out_act_grad(a.q_out_act, mean, zm);
      }
    } else {
      const int A = a.A;
      float mine = 0.f;  // lane n keeps logits[n]
      for (int n0 = 0; n0 < 2 * A; n0 += G) {
        float o[G];
        row_dots<NCH, G>(h, a.Wout[chain], W, n0, 2 * A, lane, o);
#pragma unroll
        for (int q = 0; q < G; ++q)
          if (lane == n0 + q) mine = o[q];
      }
      if (lane < 2 * A) mine += a.bout[chain][lane];
      const float mine_z = mine;
      if (lane < a.pi_out_n) mine = out_act_fwd(a.pi_out_act, mine);
      if (chain == 0 && a.pi_dact && lane < 2 * A)
        a.pi_dact[(size_t)r * 2 * A + lane] = lane < a.pi_out_n ? out_act_grad(a.pi_out_act, mine, mine_z) : 1.0f;
      const float raw = __shfl(mine, lane + A, 64);  // lane j < A: raw log-std of dim j
      float lp = 0.f;
      if (lane < A) {
        const float eps = (chain == 0 ? a.eps_new : a.eps_2)[(size_t)r * A + lane];
        const TanhGaussFwd f = tanh_gauss_fwd(mine, raw, eps, a.act_scale[lane], a.act_center[lane], a.lo_ls, a.hi_ls);
        lp = f.lp;
        float* X = chain == 0 ? a.XP : a.X2;
        X[(size_t)r * a.ldx + a.O + lane] = f.a;
        float* Xb = chain == 0 ? a.XPb : a.X2b;
        if (Xb) Xb[(size_t)r * a.ldx + a.O + lane] = f.a;
        if (chain == 0) {
          if (!a.v1_stats) { s_tanh = tanhf(mine); s_sig = f.sigma; }
          else {
            s_tanh = lane == 0 ? tanhf(mine) : 0.f;
            // logits = (mean_0..mean_{A-1}, std_0..): element 1 is mean_1, or std_0 for a one-dimensional action
            s_sig = A >= 2 ? (lane == 1 ? mine : 0.f) : f.sigma;
          }
        }
      }
      float* lg = chain == 0 ? a.logits_pi : a.logits_pit;
      if (lane < 2 * A) lg[(size_t)r * 2 * A + lane] = mine;
      lp = wave_sum(lp);
      if (lane == 0) (chain == 0 ? a.logp_new : a.logp2)[r] = lp;
    }
  }
  if (chain == 0) {
    s_tanh = wave_sum(s_tanh);
    s_sig = wave_sum(s_sig);
    if (lane == 0) { red[wave] = s_tanh; red[4 + wave] = s_sig; }
    __syncthreads();
    if (tid == 0) {
      a.part_heads[2 * blockIdx.x] = red[0] + red[1] + red[2] + red[3];
      a.part_heads[2 * blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
    }
  }
  TL_STAMP();
  TL_FLUSH(a.timeline, (int)(blockIdx.y * gridDim.x + blockIdx.x));
}

// ---------------------------------------------------------------------------------------------
// k_loss: dsac_v2.py:218-318, ONE WAVE PER SAMPLE, no block-level barriers.
//   Every global load of the wave (batch std column, the sample's scalars, the four last-hidden rows,
//   the output-layer weights, the GELU' rows) is issued at kernel entry, so the wave pays one
//   memory round trip; the per-sample scalar math is done redundantly by all lanes.
//     1  batch means of std1/std2 (each wave sums the whole column: 2B floats) -> mean_std EMA
//     2  output layers of q1_t,q2_t(obs2,act2) and q1,q2(obs,new_act)
//     3  targets / ratio / losses -> dL/d(out) of the 4 differentiated chains
//     4  dZ of the last hidden layer of those chains: (dOut . Wout) * GELU'(z)
// part_loss[row][12]: loss_q1, loss_q2, q1, q2, std1, std2, actor term, logp_new, [8] = alpha used by
// this update (row 0 only), [10],[11] = std1, std2 (consumers reduce these two with min).
// ---------------------------------------------------------------------------------------------
constexpr int kLossPart = 12;
struct LossArgs {
  const float* Hl[4];    // last hidden activations of q1_t, q2_t, q1(obs,new_act), q2(obs,new_act)
  const float* Wout[4];  // out weights: q1_target, q2_target, q1, q2   [2 x W]
  const float* bout[4];
  const float* Gl[4];    // GELU' of the last hidden layer of q1c, q2c, q1p, q2p
  float* dZl[4];         // dZ (last hidden) of q1c, q2c, q1p, q2p
  const float* qout_c[2];  // raw outs q1(obs,act), q2(obs,act)          [B x 2]
  const float* qstd_c[2];  // (std, d std / d raw) of the same             [B x 2]
  float* qout_t[2];        // raw outs of the targets (debug)              [B x 2]
  float* qout_p[2];        // raw outs q(obs,new_act)   (debug)             [B x 2]
  float* dout[4];          // dL/d(out) [B x 2]: q1c, q2c, q1p, q2p
  const float* rew; const float* done; const float* logp2; const float* logp_new;
  const float* z5; const float* z6;
  const float* log_alpha;
  float* part_loss;        // [B][kLossPart]
  float* grads_tail;       // [2] updated mean_std1/2 (re-synchronised by the gradient all-reduce)
  const DevState* st;
  int W, B;
  float inv_B;             // 1 / local batch
  float inv_Bg;            // 1 / global batch   (mean of std for the EMA; == inv_B on one GPU)
  const float* std_sums;   // {sum std1, sum std2} computed elsewhere (large B / strict data-parallel); else NULL
  int auto_alpha; float alpha_fixed, gamma, tau_b, one_minus_tau_b;
  int q_out_act;           // output activation of the critics (dsact_math.h out_act_fwd); qout_c / qstd_c hold post-activation values
  const float* qdmean[2];  // OUT_ACT_GELU: d mean_y / d z of q1 / q2(obs, act) as k_heads stored it, else nullptr
  long long* timeline;
  RideArgs ride;
};

template <int NCH>
__global__ void __launch_bounds__(kThreads) k_loss(LossArgs a) {
  if (loss_rider(a.ride)) return;
  TL_DECL
  TL_STAMP();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.B) return;  // waves are independent: no barrier anywhere in this kernel
  // ---------------- all loads ----------------
  float s1 = 0.f, s2 = 0.f;
  if (a.std_sums == nullptr) {
    // 4 independent (clamped) loads per column and trip: a plain accumulate loop compiles to
    // load -> wait -> add per element, i.e. one memory round trip per 64 samples
    for (int i0 = 0; i0 < a.B; i0 += 256) {
      float v1[4], v2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = i0 + u * 64 + lane;
        const int rc = rr < a.B ? rr : 0;
        v1[u] = a.qstd_c[0][2 * rc]; v2[u] = a.qstd_c[1][2 * rc];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = i0 + u * 64 + lane < a.B;
        s1 += ok ? v1[u] : 0.f; s2 += ok ? v2[u] : 0.f;
      }
    }
  } else if (lane == 0) { s1 = a.std_sums[0]; s2 = a.std_sums[1]; }
  const float q1 = a.qout_c[0][2 * r], q2 = a.qout_c[1][2 * r];
  const float std1 = a.qstd_c[0][2 * r], sg1 = a.qstd_c[0][2 * r + 1];
  const float std2 = a.qstd_c[1][2 * r], sg2 = a.qstd_c[1][2 * r + 1];
  const float in_z5 = a.z5[r], in_z6 = a.z6[r], rew = a.rew[r], in_done = a.done[r];
  const float lp2 = a.logp2[r], lpn = a.logp_new[r];
  const float la = a.log_alpha[0];   // unconditional: a load inside a branch is drained at the join
  const float ms1_old = a.st->ms1, ms2_old = a.st->ms2;
  const int ms_init = a.st->ms_init;
  float bo[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c) { bo[c][0] = a.bout[c][0]; bo[c][1] = a.bout[c][1]; }
  f32x4 h[4][NCH], wv[4][2][NCH], gv[4][NCH];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    row_load<NCH>(a.Hl[c] + (size_t)r * a.W, a.W, lane, h[c]);
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int k = q * 256 + lane * 4;
      const int kc = k < a.W ? k : 0;
      wv[c][0][q] = *(const f32x4u*)(a.Wout[c] + kc);
      wv[c][1][q] = *(const f32x4u*)(a.Wout[c] + a.W + kc);
      gv[c][q] = *(const f32x4u*)(a.Gl[c] + (size_t)r * a.W + kc);
    }
  }
  TL_STAMP();  // 1: loads issued
#pragma unroll
  for (int c = 0; c < 4; ++c) row_mask<NCH>(h[c], a.W, lane);
  // ---------------- 2: output layers ----------------
  float o[4][2], zo[4][2];   // (zo: pre-activation, for the one output activation whose derivative needs it)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < NCH; ++q) {
        s += h[c][q].x * wv[c][j][q].x; s += h[c][q].y * wv[c][j][q].y;
        s += h[c][q].z * wv[c][j][q].z; s += h[c][q].w * wv[c][j][q].w;
      }
      zo[c][j] = wave_sum(s) + bo[c][j];
      o[c][j] = out_act_fwd(a.q_out_act, zo[c][j]);
    }
  // ---------------- 1: mean_std EMA ----------------
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  const float m1 = s1 * a.inv_Bg, m2 = s2 * a.inv_Bg;
  float ms1, ms2;
  if (!ms_init) { ms1 = m1; ms2 = m2; }
  else {
    ms1 = a.one_minus_tau_b * ms1_old + a.tau_b * m1;  // Python double (1 - tau_b) cast to fp32 by the tensor multiply
    ms2 = a.one_minus_tau_b * ms2_old + a.tau_b * m2;
  }
  const float alpha = a.auto_alpha ? expf(la) : a.alpha_fixed;
  TL_STAMP();  // 2: reductions done
  // ---------------- 3: per-sample math (uniform across the wave) ----------------
  const float q1n = o[0][0], std1n = softplus(o[0][1]);
  const float q2n = o[1][0], std2n = softplus(o[1][1]);
  const float qn = fminf(q1n, q2n);
  const float z5 = clampf(in_z5, -3.f, 3.f), z6 = clampf(in_z6, -3.f, 3.f);
  const float qs = (q1n < q2n) ? (q1n + z5 * std1n) : (q2n + z6 * std2n);
  const float nd = 1.0f - in_done;
  const float tq = rew + nd * a.gamma * (qn - alpha * lp2);
  const float tqs = rew + nd * a.gamma * (qs - alpha * lp2);
  const CriticTerm c1 = critic_term(q1, std1, ms1, tq, tqs);
  const CriticTerm c2 = critic_term(q2, std2, ms2, tq, tqs);
  float dv[8];
  // (d / d pre-activation output: the std column's factor is folded into sg by k_heads, the mean column's is formed here)
  dv[0] = c1.dq * a.inv_B * (a.qdmean[0] ? a.qdmean[0][r] : out_act_grad_y(a.q_out_act, q1));
  dv[1] = c1.dstd * a.inv_B * sg1;
  dv[2] = c2.dq * a.inv_B * (a.qdmean[1] ? a.qdmean[1][r] : out_act_grad_y(a.q_out_act, q2));
  dv[3] = c2.dstd * a.inv_B * sg2;
  // actor: mean(alpha*logp_new - min(q1p, q2p)); torch.min ties split the gradient evenly
  const float q1p = o[2][0], q2p = o[3][0];
  const float w1 = q1p < q2p ? 1.0f : (q1p > q2p ? 0.0f : 0.5f);
  dv[4] = -w1 * a.inv_B * out_act_grad(a.q_out_act, q1p, zo[2][0]); dv[5] = 0.0f;
  dv[6] = -(1.0f - w1) * a.inv_B * out_act_grad(a.q_out_act, q2p, zo[3][0]); dv[7] = 0.0f;
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      a.dout[c][2 * r] = dv[2 * c]; a.dout[c][2 * r + 1] = dv[2 * c + 1];
      float* dbg = c < 2 ? a.qout_t[c] : a.qout_p[c - 2];
      dbg[2 * r] = o[c][0]; dbg[2 * r + 1] = o[c][1];
    }
    float* pl = a.part_loss + (size_t)r * kLossPart;
    pl[0] = c1.loss; pl[1] = c2.loss; pl[2] = q1; pl[3] = q2; pl[4] = std1; pl[5] = std2;
    pl[6] = alpha * lpn - fminf(q1p, q2p);
    pl[7] = lpn;
    pl[8] = r == 0 ? alpha : 0.0f;  // tb_info reports the alpha the losses used
    pl[9] = 0.0f; pl[10] = std1; pl[11] = std2;
    if (r == 0) { a.grads_tail[0] = ms1; a.grads_tail[1] = ms2; }
  }
  TL_STAMP();  // 3: per-sample math done
  // ---------------- 4: dZ of the last hidden layer ----------------
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float d0 = dv[2 * c], d1 = dv[2 * c + 1];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      const int k = q * 256 + lane * 4;
      if (k < a.W) {
        // chains q1c/q1p differentiate through q1's output layer (weights of chain 2), q2c/q2p through q2's (3)
        const f32x4 w0 = wv[2 + (c & 1)][0][q], w1v = wv[2 + (c & 1)][1][q];
        f32x4 ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) ov[e] = (d0 * w0[e] + d1 * w1v[e]) * gv[c][q][e];
        float* dz = a.dZl[c] + (size_t)r * a.W + k;
        if (k + 3 < a.W) *(f32x4u*)dz = ov;
        else for (int e = 0; e < 4 && k + e < a.W; ++e) dz[e] = ov[e];
      }
    }
  }
  TL_STAMP();  // 4: dZ written
  TL_FLUSH(a.timeline, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_loss_v1: DSAC_V1 (dsac_v1.py:194-253), one wave per sample like k_loss. One critic:
//   q_next_sample = mean_t + clamp(z,+-3) * std_t          (q_target(obs2, act2), dsac_v1.py:184-192)
//   target_q = r + (1-d) gamma (q_next_sample - alpha logp2); target_q_bound = q + clamp(target_q - q, +-TD_bound)
//   L_q = mean( -(target_q - q)/(std^2 + 0.1) * q - ((q - target_q_bound)^2 - std^2)/(std^3 + 0.1) * std )
//         with detached coefficients  ->  dL/dq, dL/dstd are the coefficients themselves / B
//   actor: mean(alpha logp_new - q(obs,new_act))  ->  dL/dq_pi = -1/B
// chains: [0] q_target(obs2,act2) (forward only), [1] q(obs,new_act); differentiated: q(obs,act) via Gl[0]/dZl[0],
// q(obs,new_act) via Gl[1]/dZl[1]. part_loss row: [2] q, [4] std, [6] actor term, [7] logp_new, [8] alpha.
// ---------------------------------------------------------------------------------------------
struct LossV1Args {
  const float* Hl[2];    // last hidden activations of q_target(obs2,act2), q(obs,new_act)
  const float* Wout[2];  // out weights of q_target, q   [2 x W]
  const float* bout[2];
  const float* Gl[2];    // GELU' of the last hidden layer of q(obs,act), q(obs,new_act)
  float* dZl[2];
  const float* qout_c;   // raw outs of q(obs,act)  [B x 2]
  const float* qstd_c;   // (std, d std / d raw)    [B x 2]
  float* qout_t; float* qout_p;   // debug
  float* dout[2];        // dL/d(out) [B x 2] of q(obs,act), q(obs,new_act)
  const float* rew; const float* done; const float* logp2; const float* logp_new; const float* z_t;
  const float* log_alpha;
  float* part_loss; float* grads_tail;
  int W, B;
  float inv_B;
  int auto_alpha; float alpha_fixed, gamma, td_bound;
  int bound;             // dsac_v1.py:217: 1 = variance-weighted pseudo-loss with the clipped target; 0 = -Normal(q, std).log_prob(target)
  RideArgs ride;
};

template <int NCH>
__global__ void __launch_bounds__(kThreads) k_loss_v1(LossV1Args a) {
  if (loss_rider(a.ride)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.B) return;
  const float q = a.qout_c[2 * r], std = a.qstd_c[2 * r], sg = a.qstd_c[2 * r + 1];
  const float in_z = a.z_t[r], rew = a.rew[r], in_done = a.done[r], lp2 = a.logp2[r], lpn = a.logp_new[r];
  const float la = a.log_alpha[0];
  float bo[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c) { bo[c][0] = a.bout[c][0]; bo[c][1] = a.bout[c][1]; }
  f32x4 h[2][NCH], wv[2][2][NCH], gv[2][NCH];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    row_load<NCH>(a.Hl[c] + (size_t)r * a.W, a.W, lane, h[c]);
#pragma unroll
    for (int qq = 0; qq < NCH; ++qq) {
      const int k = qq * 256 + lane * 4;
      const int kc = k < a.W ? k : 0;
      wv[c][0][qq] = *(const f32x4u*)(a.Wout[c] + kc);
      wv[c][1][qq] = *(const f32x4u*)(a.Wout[c] + a.W + kc);
      gv[c][qq] = *(const f32x4u*)(a.Gl[c] + (size_t)r * a.W + kc);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) row_mask<NCH>(h[c], a.W, lane);
  float o[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int qq = 0; qq < NCH; ++qq) {
        s += h[c][qq].x * wv[c][j][qq].x; s += h[c][qq].y * wv[c][j][qq].y;
        s += h[c][qq].z * wv[c][j][qq].z; s += h[c][qq].w * wv[c][j][qq].w;
      }
      o[c][j] = wave_sum(s) + bo[c][j];
    }
  const float alpha = a.auto_alpha ? expf(la) : a.alpha_fixed;
  const float qn = o[0][0], stdn = softplus(o[0][1]);
  const float qs = qn + clampf(in_z, -3.f, 3.f) * stdn;
  const float tq = rew + (1.0f - in_done) * a.gamma * (qs - alpha * lp2);
  const float tqb = q + clampf(tq - q, -a.td_bound, a.td_bound);
  const float sd = fmaxf(std, 0.0f);
  float dq = -(tq - q) / (sd * sd + 0.1f);
  const float e = q - tqb;
  float dstd = -((e * e - sd * sd) / (sd * sd * sd + 0.1f));
  float loss_row = dq * q + dstd * std;
  if (!a.bound) {
    // dsac_v1.py:227-228: -Normal(q, std).log_prob(tq) = (tq - q)^2 / (2 std^2) + log std + log sqrt(2 pi)
    const float d = tq - q, var = std * std;
    dq = -d / var;
    dstd = 1.0f / std - (d * d) / (var * std);
    loss_row = (d * d) / (2.0f * var) + logf(std) + kLogSqrt2Pi;
  }
  float dv[4];
  dv[0] = dq * a.inv_B;
  dv[1] = dstd * a.inv_B * sg;
  dv[2] = -a.inv_B; dv[3] = 0.0f;
  const float q_pi = o[1][0];
  if (lane == 0) {
    a.dout[0][2 * r] = dv[0]; a.dout[0][2 * r + 1] = dv[1];
    a.dout[1][2 * r] = dv[2]; a.dout[1][2 * r + 1] = dv[3];
    a.qout_t[2 * r] = o[0][0]; a.qout_t[2 * r + 1] = o[0][1];
    a.qout_p[2 * r] = o[1][0]; a.qout_p[2 * r + 1] = o[1][1];
    float* pl = a.part_loss + (size_t)r * kLossPart;
    pl[0] = loss_row; pl[1] = 0.f; pl[2] = q; pl[3] = 0.f; pl[4] = std; pl[5] = 0.f;
    pl[6] = alpha * lpn - q_pi;
    pl[7] = lpn;
    pl[8] = r == 0 ? alpha : 0.0f;
    pl[9] = 0.0f; pl[10] = std; pl[11] = std;
    if (r == 0) { a.grads_tail[0] = 0.f; a.grads_tail[1] = 0.f; }
  }
  // dZ of the last hidden layer of q(obs,act) and q(obs,new_act): both through q's output layer (chain 1's weights)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float d0 = dv[2 * c], d1 = dv[2 * c + 1];
#pragma unroll
    for (int qq = 0; qq < NCH; ++qq) {
      const int k = qq * 256 + lane * 4;
      if (k < a.W) {
        const f32x4 w0 = wv[1][0][qq], w1v = wv[1][1][qq];
        f32x4 ov;
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) ov[e2] = (d0 * w0[e2] + d1 * w1v[e2]) * gv[c][qq][e2];
        float* dz = a.dZl[c] + (size_t)r * a.W + k;
        if (k + 3 < a.W) *(f32x4u*)dz = ov;
        else for (int e2 = 0; e2 < 4 && k + e2 < a.W; ++e2) dz[e2] = ov[e2];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_heads_bwd: actor path between the Q nets' first layer and the policy's last hidden layer.
//   dA[r][j]   = sum_k dZ1_q1p[r][k] W1_q1[k][O+j] + sum_k dZ1_q2p[r][k] W1_q2[k][O+j]
//                (lanes over k; rows of the transposed action columns are fetched 8 at a time --
//                 independent coalesced loads -- and reduced with DPP)
//   (dmu,draw) = tanh-Gaussian rsample backward with dL/dlogp = alpha/B     (App. A.3)
//   dZ_pi_last = ((dmu|draw) . Wout_pi) * GELU'(z_last)
// one wave per row. Block 0 / wave 0 also finalises the alpha gradient (dsac_v2.py:312-318):
//   d loss_alpha / d log_alpha = -mean(logp_new + target_entropy)
// ---------------------------------------------------------------------------------------------
struct HeadsBwdArgs {
  const float* dZ1[2];     // [B x W0] first-hidden dZ of q1(obs,new_act), q2(obs,new_act)
  const float* W1aT[2];    // [32][W0] transposed, zero padded action columns of q1 / q2's first layer
  int W0;
  const float* logits_pi;  // [B x 2A]
  const float* eps_new;
  const float* log_alpha;
  const float* Wout_pi;    // [2A x WL]
  const float* G_pi;       // GELU' of the policy's last hidden layer [B x WL]
  float* dZ_pi;            // [B x WL]
  float* dout_pi;          // [B x 2A]
  float* d_new_act;        // [B x A] (debug / parity)
  int WL, B, O, A;
  float inv_B; int auto_alpha; float alpha_fixed;
  const float* act_scale; float lo_ls, hi_ls;
  const float* part_loss; int n_part; float target_entropy; float* grad_log_alpha;
  int pi_out_act, pi_out_n;   // policy output activation and the outputs it applies to (HeadsArgs); logits_pi holds post-activation values
  const float* pi_dact;       // OUT_ACT_GELU: d logit_y / d z [B x 2A] as k_heads stored it, else nullptr
  long long* timeline;
  // ride-along weight-gradient tiles of the critics (blocks >= n_row_blocks): this launch has only B/4 row blocks, and
  // the critics' dW is complete (and their weights free) as soon as the critic backward is
  const GemmProb* extra; int n_extra; int n_row_blocks;
  FusedOpt fo;
};

template <int NCH>
__global__ void __launch_bounds__(kThreads) k_heads_bwd(HeadsBwdArgs a) {
  __shared__ float sh_dout[4][64];
  extern __shared__ __attribute__((aligned(16))) float tile_lds[];
  if ((int)blockIdx.x >= a.n_row_blocks) {
    const GemmProb g = a.extra[blockIdx.x - a.n_row_blocks];
    run_tile<true, true, EPI_STORE>(g, g.tiles_n, g.tile_end, tile_lds, nullptr, 0, &a.fo);
    return;
  }
  TL_DECL
  TL_STAMP();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (blockIdx.x == 0 && wave == 0) {
    float s = 0.f;
    for (int i0 = 0; i0 < a.n_part; i0 += 256) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 64 + lane;
        v[u] = a.part_loss[(size_t)(i < a.n_part ? i : 0) * kLossPart + 7];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (i0 + u * 64 + lane < a.n_part) ? v[u] : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) a.grad_log_alpha[0] = a.auto_alpha ? -(s * a.inv_B + a.target_entropy) : 0.0f;
  }
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.B) return;  // whole wave exits together; no block-level barrier below
  const int A = a.A;
  float mu = 0.f, raw = 0.f, eps = 0.f, scale = 1.f, dA = 0.f;
  if (lane < A) {
    mu = a.logits_pi[(size_t)r * 2 * A + lane];
    raw = a.logits_pi[(size_t)r * 2 * A + A + lane];
    eps = a.eps_new[(size_t)r * A + lane];
    scale = a.act_scale[lane];
  }
  {
    // this lane's slice of the two dZ1 rows (4 hidden units per 256-chunk), zero beyond the row
    f32x4 d1[4], d2[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = c * 256 + lane * 4;
      const int kc = k < a.W0 ? k : 0;
      d1[c] = *(const f32x4u*)(a.dZ1[0] + (size_t)r * a.W0 + kc);   // raw: zeroed beyond the row at use time
      d2[c] = *(const f32x4u*)(a.dZ1[1] + (size_t)r * a.W0 + kc);
    }
    const int nch0 = (a.W0 + 255) >> 8;
    // groups of 12 action dimensions: 24 independent row loads per chunk, then 12 DPP reductions.
    // Rows j >= A of the transposed copy are zero (repack), so no masking is needed up to 32
    // (the last group clamps its row index).
    constexpr int JG = 12;
    for (int j0 = 0; j0 < A; j0 += JG) {
      float sacc[JG];
#pragma unroll
      for (int q = 0; q < JG; ++q) sacc[q] = 0.f;
      for (int c = 0; c < nch0; ++c) {
        const int k = c * 256 + lane * 4;
        const int kc = k < a.W0 ? k : 0;   // lanes beyond the row multiply by d = 0
        f32x4 w1[JG], w2[JG];
#pragma unroll
        for (int q = 0; q < JG; ++q) {
          const int j = j0 + q < 32 ? j0 + q : 31;
          w1[q] = *(const f32x4u*)(a.W1aT[0] + (size_t)j * a.W0 + kc);
          w2[q] = *(const f32x4u*)(a.W1aT[1] + (size_t)j * a.W0 + kc);
        }
        f32x4 e1 = c == 0 ? d1[0] : c == 1 ? d1[1] : c == 2 ? d1[2] : d1[3];
        f32x4 e2 = c == 0 ? d2[0] : c == 1 ? d2[1] : c == 2 ? d2[2] : d2[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) if (k + e >= a.W0) { e1[e] = 0.f; e2[e] = 0.f; }
#pragma unroll
        for (int q = 0; q < JG; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[q] += e1[e] * w1[q][e] + e2[e] * w2[q][e];
      }
#pragma unroll
      for (int q = 0; q < JG; ++q) {
        const float t = wave_sum(sacc[q]);
        if (lane == j0 + q) dA = t;
      }
    }
  }
  // prefetch this lane's GELU' values of the row (used at the very end)
  f32x4 gv[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 256 + lane * 4;
    gv[c] = *(const f32x4u*)(a.G_pi + (size_t)r * a.WL + (k < a.WL ? k : 0));
  }
  const float alpha = a.auto_alpha ? expf(a.log_alpha[0]) : a.alpha_fixed;
  TL_STAMP();  // 1: dL/d new_act formed
  float dmu = 0.f, draw = 0.f;
  if (lane < A) {
    tanh_gauss_bwd(mu, raw, eps, scale, a.lo_ls, a.hi_ls, dA, alpha * a.inv_B, dmu, draw);
    if (a.pi_dact) { dmu *= a.pi_dact[(size_t)r * 2 * A + lane]; draw *= a.pi_dact[(size_t)r * 2 * A + A + lane]; }
    else {
      dmu *= out_act_grad_y(a.pi_out_act, mu);
      if (A + lane < a.pi_out_n) draw *= out_act_grad_y(a.pi_out_act, raw);
    }
    a.dout_pi[(size_t)r * 2 * A + lane] = dmu;
    a.dout_pi[(size_t)r * 2 * A + A + lane] = draw;
    a.d_new_act[(size_t)r * A + lane] = dA;
    sh_dout[wave][lane] = dmu;
    sh_dout[wave][A + lane] = draw;
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  TL_STAMP();  // 2: rsample backward done
  // dZ of the policy's last hidden layer: lane owns 4 consecutive hidden units per 256-chunk
  f32x4 s[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
  for (int n = 0; n < 2 * A; ++n) {
    const float d = sh_dout[wave][n];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int k = c * 256 + lane * 4;
      const f32x4 wv = *(const f32x4u*)(a.Wout_pi + (size_t)n * a.WL + (k < a.WL ? k : 0));
      s[c] += d * wv;
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int k = c * 256 + lane * 4;
    if (k < a.WL) {
      float* dz = a.dZ_pi + (size_t)r * a.WL + k;
      const f32x4 o = s[c] * gv[c];
      if (k + 3 < a.WL) *(f32x4u*)dz = o;
      else for (int q = 0; q < 4 && k + q < a.WL; ++q) dz[q] = o[q];
    }
  }
  TL_STAMP();  // 3: dZ_pi written
  TL_FLUSH(a.timeline, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// k_adam: DSAC_V2.__update (dsac_v2.py:320-347) as one streaming pass over the flat arenas.
// ---------------------------------------------------------------------------------------------
struct AdamArgs {
  float* p; float* tgt; float* m; float* v; const float* g;
  long long n_q2;      // floats in q1|q2
  long long n_online3; // floats in q1|q2|policy
  long long n_total;   // + log_alpha
  DevState* st;
  float b1w, beta2, b2w, eps;
  float polyak, one_minus_polyak;
  int auto_alpha;
  int commit_ms;       // 1: take mean_std from g[n_total..n_total+1]
  // split-K weight gradients (batch > 448): the gradient of element i < n_total-1 is the sum of n_part chunk partials
  // (log_alpha's gradient and the mean_std tail always come from g); NULL: gradients are in g
  const float* part; long long part_stride; int n_part;
  long long direct_lo[3], direct_hi[3];   // conv parameters of q1 / q2 / policy: their gradients are always in g
};

__device__ __forceinline__ bool adam_direct(const AdamArgs& a, long long i) {
  return i >= a.n_total - 1 || (i >= a.direct_lo[0] && i < a.direct_hi[0]) || (i >= a.direct_lo[1] && i < a.direct_hi[1]) ||
         (i >= a.direct_lo[2] && i < a.direct_hi[2]);
}
__device__ __forceinline__ f32x4 adam_grad4(const AdamArgs& a, long long base) {
  if (a.part == nullptr) return *(const f32x4*)(a.g + base);
  f32x4 s = *(const f32x4*)(a.part + base);
  for (int c = 1; c < a.n_part; ++c) s += *(const f32x4*)(a.part + c * a.part_stride + base);
#pragma unroll
  for (int e = 0; e < 4; ++e) if (adam_direct(a, base + e)) s[e] = a.g[base + e];
  return s;
}
__device__ __forceinline__ float adam_grad1(const AdamArgs& a, long long i) {
  if (a.part == nullptr || adam_direct(a, i)) return a.g[i];
  float s = a.part[i];
  for (int c = 1; c < a.n_part; ++c) s += a.part[c * a.part_stride + i];
  return s;
}

__device__ __forceinline__ void adam_classify(const AdamArgs& a, const DevState& st, long long base, bool delayed,
                                              bool (&upd)[4], float (&ss)[4], float (&bc2)[4], bool& any) {
  any = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long long i = base + e;
    if (i < a.n_q2) { upd[e] = true; ss[e] = st.ss_q; bc2[e] = st.bc2_q; }
    else if (i < a.n_online3) { upd[e] = delayed; ss[e] = st.ss_pi; bc2[e] = st.bc2_pi; }
    else { upd[e] = delayed && a.auto_alpha && i < a.n_total; ss[e] = st.ss_alpha; bc2[e] = st.bc2_alpha; }
    any = any || upd[e];
  }
}

#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_adam(AdamArgs a) {
  const DevState st = *a.st;
  const long long n4 = (a.n_total + 3) >> 2;
  const bool delayed = st.do_delayed != 0;
  // each thread owns two float4 groups (i4 and i4 + half): all their loads are issued before any math
  const long long half = (n4 + 1) >> 1;
  const long long i0 = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (i0 < half) {
    long long base[2] = {i0 << 2, (i0 + half) << 2};
    bool upd[2][4], any[2], vec[2], tvec[2];
    float ss[2][4], bc2[2][4];
    f32x4 p[2], m[2], v[2], g[2], t[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const bool exists = (base[u] >> 2) < n4 && (u == 0 || i0 + half < n4);
      if (exists) adam_classify(a, st, base[u], delayed, upd[u], ss[u], bc2[u], any[u]); else any[u] = false;
      vec[u] = any[u] && base[u] + 3 < a.n_total;
      tvec[u] = vec[u] && delayed && base[u] + 3 < a.n_online3;
      if (vec[u]) {
        p[u] = *(const f32x4*)(a.p + base[u]); m[u] = *(const f32x4*)(a.m + base[u]);
        v[u] = *(const f32x4*)(a.v + base[u]); g[u] = adam_grad4(a, base[u]);
        if (tvec[u]) t[u] = *(const f32x4*)(a.tgt + base[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!any[u]) continue;  // policy / alpha segments on the off iterations of the delayed update
      if (vec[u]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (upd[u][e]) {
            float pe = p[u][e], me = m[u][e], ve = v[u][e];
            adam_update(pe, me, ve, g[u][e], a.b1w, a.beta2, a.b2w, ss[u][e], bc2[u][e], a.eps);
            p[u][e] = pe; m[u][e] = me; v[u][e] = ve;
          }
          if (tvec[u]) t[u][e] = polyak_update(t[u][e], p[u][e], a.polyak, a.one_minus_polyak);
          else if (delayed && base[u] + e < a.n_online3)
            a.tgt[base[u] + e] = polyak_update(a.tgt[base[u] + e], p[u][e], a.polyak, a.one_minus_polyak);
        }
        *(f32x4*)(a.p + base[u]) = p[u]; *(f32x4*)(a.m + base[u]) = m[u]; *(f32x4*)(a.v + base[u]) = v[u];
        if (tvec[u]) *(f32x4*)(a.tgt + base[u]) = t[u];
      } else {
        for (int e = 0; e < 4 && base[u] + e < a.n_total; ++e) {
          const long long i = base[u] + e;
          float pe = a.p[i];
          if (upd[u][e]) {
            float me = a.m[i], ve = a.v[i];
            adam_update(pe, me, ve, adam_grad1(a, i), a.b1w, a.beta2, a.b2w, ss[u][e], bc2[u][e], a.eps);
            a.p[i] = pe; a.m[i] = me; a.v[i] = ve;
          }
          if (delayed && i < a.n_online3) a.tgt[i] = polyak_update(a.tgt[i], pe, a.polyak, a.one_minus_polyak);
        }
      }
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (a.commit_ms) { a.st->ms1 = a.g[a.n_total]; a.st->ms2 = a.g[a.n_total + 1]; a.st->ms_init = 1; }
    a.st->it_next = st.it_cur + 1;
    a.st->seq_next = st.seq_next + 1;
    a.st->tag_seq = st.tag_seq + 1;
  }
}
#endif

// split-K weight gradients (batch > 448): every 256-sample chunk of the batch writes its own partial gradient arena;
// this pass adds them in chunk order into the gradient arena [0, n) (log_alpha's gradient and the mean_std tail are
// produced elsewhere and left alone)
struct SumPartsArgs { const float* part; long long stride; int n_part; float* g; long long n; };
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_sum_parts(SumPartsArgs a) {
  const long long i4 = ((long long)blockIdx.x * kThreads + threadIdx.x) * 4;
  if (i4 >= a.n) return;
  if (i4 + 3 < a.n) {
    f32x4 s = *(const f32x4u*)(a.part + i4);
    for (int c = 1; c < a.n_part; ++c) s += *(const f32x4u*)(a.part + c * a.stride + i4);
    *(f32x4u*)(a.g + i4) = s;
  } else {
    for (long long i = i4; i < a.n; ++i) {
      float s = a.part[i];
      for (int c = 1; c < a.n_part; ++c) s += a.part[c * a.stride + i];
      a.g[i] = s;
    }
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// k_stats: the 14 numeric tb_info entries (dsac_v2.py:188-202) from the partial sums; launched
// only when the host asks (trainer logs every log_save_interval iterations).
// ---------------------------------------------------------------------------------------------
// dst[i] = src[idx[i]]  (dsact_read_batch: the sampled rows' logp column, one launch + one copy)
struct TakeArgs { const float* src; const int* idx; float* dst; int n; };
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void k_take(TakeArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) a.dst[i] = a.src[a.idx[i]];
}
#endif

struct StatsArgs {
  const float* part_loss; int n_loss; const float* part_heads; int n_heads;
  const float* log_alpha; const DevState* st; float inv_B; float inv_BA; int auto_alpha; float alpha_fixed;
  float* out;  // [16]
  const float* ms_tail;   // nullptr, or mean_std1/2 of the gradient that is pending (not yet committed to DevState)
  const int* spin_timeout;   // merged forward launch: a consumer gave up waiting for its producers -> every statistic reads NaN
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void k_stats(StatsArgs a) {
  const int lane = threadIdx.x;
  float s[kLossPart];
  for (int k = 0; k < kLossPart; ++k) s[k] = k < 10 ? 0.f : INFINITY;
  for (int i = lane; i < a.n_loss; i += 64)
    for (int k = 0; k < kLossPart; ++k) {
      const float v = a.part_loss[i * kLossPart + k];
      s[k] = k < 10 ? s[k] + v : fminf(s[k], v);
    }
  for (int k = 0; k < kLossPart; ++k) s[k] = k < 10 ? wave_sum(s[k]) : wave_min(s[k]);
  float h0 = 0.f, h1 = 0.f;
  for (int i = lane; i < a.n_heads; i += 64) { h0 += a.part_heads[2 * i]; h1 += a.part_heads[2 * i + 1]; }
  h0 = wave_sum(h0); h1 = wave_sum(h1);
  if (lane == 0) {
    float* o = a.out;
    o[0] = s[2] * a.inv_B; o[1] = s[3] * a.inv_B; o[2] = s[4] * a.inv_B; o[3] = s[5] * a.inv_B;
    o[4] = s[10]; o[5] = s[11];
    o[6] = s[6] * a.inv_B;
    o[7] = s[0] * a.inv_B + s[1] * a.inv_B;
    o[8] = h0 * a.inv_BA; o[9] = h1 * a.inv_BA;
    o[10] = -(s[7] * a.inv_B);
    o[11] = s[8];
    o[12] = a.ms_tail ? a.ms_tail[0] : a.st->ms1; o[13] = a.ms_tail ? a.ms_tail[1] : a.st->ms2;
    o[14] = (float)a.st->it_cur; o[15] = 0.f;
    if (a.spin_timeout && *a.spin_timeout)
      for (int k = 0; k < 14; ++k) o[k] = NAN;
  }
}
#endif

// strict data-parallel mode: local {sum std1, sum std2} for the pre-loss all-reduce
struct StdSumArgs { const float* qstd_c[2]; int B; float* out; };
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_std_sums(StdSumArgs a) {
  __shared__ float red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s1 = 0.f, s2 = 0.f;
  for (int r = tid; r < a.B; r += kThreads) {
    s1 += a.qstd_c[0][2 * r];
    s2 += a.qstd_c[1][2 * r];
  }
  s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
  __syncthreads();
  if (tid == 0) { a.out[0] = red[0] + red[1] + red[2] + red[3]; a.out[1] = red[4] + red[5] + red[6] + red[7]; }
}
#endif

// policy head only (sampler / evaluator feed): logits (mean | std) as StochaPolicy.forward returns
struct PolicyOutArgs { const float* H; const float* Wout; const float* bout; int W, n, A; float lo_ls, hi_ls; float* out; int out_act, out_n; };
template <int NCH>
__global__ void __launch_bounds__(kThreads) k_policy_out(PolicyOutArgs a) {
  constexpr int G = NCH == 1 ? 12 : (NCH == 2 ? 8 : 4);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= a.n) return;
  f32x4 h[NCH];
  row_load<NCH>(a.H + (size_t)r * a.W, a.W, lane, h);
  for (int n0 = 0; n0 < 2 * a.A; n0 += G) {
    float o[G];
    row_dots<NCH, G>(h, a.Wout, a.W, n0, 2 * a.A, lane, o);
#pragma unroll
    for (int q = 0; q < G; ++q) {
      const int n = n0 + q;
      if (lane == 0 && n < 2 * a.A) {
        float v = o[q] + a.bout[n];
        if (n < a.out_n) v = out_act_fwd(a.out_act, v);
        if (n >= a.A) v = expf(clampf(v, a.lo_ls, a.hi_ls));
        a.out[(size_t)r * 2 * a.A + n] = v;
      }
    }
  }
}

}  // namespace dsact
