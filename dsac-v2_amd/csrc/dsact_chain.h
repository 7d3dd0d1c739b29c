// dsact_chain.h -- row-slice fused MLP chains of the DSAC-T update (gfx950, wave64, fp32 MFMA).
//
// Forward layers and the hidden-layer backward  dZ[l-1] = (dZ[l] W_l) * gelu'(z[l-1])  are independent per batch
// row, so ONE workgroup can take a 16-row slice of the minibatch through a whole network chain: first layer ->
// hidden layers -> output layer -> tanh-Gaussian rsample / softplus epilogue (forward), or loss -> output-layer
// backward -> hidden layers -> dL/d(action) (backward), with the activations in LDS and no inter-workgroup
// synchronisation. One update becomes 5 launches (was 14):
//   k_chain_fwd  A   policy(obs), policy_target(obs2), q1/q2(obs,act), + the obs2 part of q1_t/q2_t's first layer
//   k_chain_fwd  B   q1_t/q2_t(obs2,act2), q1/q2(obs,new_act): first layer = saved obs part + K=A action part
//   k_chain_bwd_q    DSAC-T loss (dsac_v2.py:218-318) + dZ chains of q1c,q2c,q1p,q2p + dL/d new_act
//   k_chain_bwd_pi   rsample backward + policy dZ chain   (+ the critics' dW/Adam tiles on the idle CUs)
//   k_stage_table    policy dW/Adam tiles + close of the update
// Reference math: networks/mlp.py:79-127, utils/act_distribution_cls.py:44-54, dsac_v2.py:150-318.
//
// What bounds a chain workgroup (scripts/ubench/slice_gemm.hip, cu_stream.hip; profiles/r02_ubench_*.txt):
//   * 16 rows x 256 outputs x K=256 is 1024 v_mfma_f32_16x16x4_f32 = 8192 cycles per SIMD = 3.4 us: the floor.
//   * every weight byte is used by exactly ONE wave (the wave that owns those output features), so weights go
//     L2 -> registers, never through LDS. A wave-load must be ONE contiguous KB: the row-major pattern the MFMA
//     fragment implies (16 rows x 64 B) runs at 14 B/clk per CU, a contiguous KB at 64 B/clk, and the chain needs 32.
//     Hence the FRAGMENT-MAJOR ("packed") weight copies below, kept fresh by the Adam tiles that own the parameters.
//   * f32 MFMA and the f32 VALU are the same lanes (a VALU-FMA wave beside an MFMA wave on one SIMD: time = sum).
//   * loads are issued between MFMA groups (a wave issues in order) and run kDc-1 chunks ahead, ACROSS layer
//     boundaries: the stream of weight chunks never drains at an epilogue or barrier (the barrier is raw s_barrier:
//     __syncthreads() would wait for vmcnt(0)).
#pragma once
#include "dsact_kernels.h"

namespace dsact {

constexpr int kChMaxL = 4;     // == DSACT_MAX_HIDDEN_LAYERS
constexpr int kDc = 4;         // weight chunk buffers per tile (a chunk = 16 k of one 16-row tile = 1 KB per wave-load)
constexpr int kChRows = 16;    // batch rows per chain workgroup

// phase stamps of the chain kernels (instrumented builds, -DDSACT_TIMELINE): [block][16] shader-clock values
#ifdef DSACT_TIMELINE
#define CTL(buf, k) do { if ((buf) && threadIdx.x == 0 && blockIdx.x < 256) (buf)[blockIdx.x * 16 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define CTL(buf, k) do {} while (0)
#endif

// ---- the weight-chunk stream ---------------------------------------------------------------------------------
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT> struct WStream { f32x4 b[kDc][NT]; };
// wave-uniform base of each of the wave's NT tiles inside a packed tensor (chunk c of tile t: p[t] + c*256 + lane*4)
template <int NT> struct WPtr { const float* p[NT]; };

template <int NT>
__device__ __forceinline__ WPtr<NT> wptr(const float* base, int C, int tile0) {
  WPtr<NT> r;
#pragma unroll
  for (int t = 0; t < NT; ++t) r.p[t] = base + (size_t)(tile0 + t) * C * 256;
  return r;
}

// first kDc-1 chunks of a stream that starts at chunk c_lo of `cur`
template <int NT>
__device__ __forceinline__ void stream_prologue(WStream<NT>& ws, const WPtr<NT>& cur, int c_lo, int lane4) {
#pragma unroll
  for (int u = 0; u < kDc - 1; ++u)
#pragma unroll
    for (int t = 0; t < NT; ++t) ws.b[u][t] = gload4(cur.p[t] + (size_t)(c_lo + u) * 256 + lane4);
}

// acc[t] += Xs[16 x 16(c_hi-c_lo)] . W_tile_t^T over chunks [c_lo, c_hi) of `cur` ((c_hi - c_lo) % kDc == 0).
// While chunk c is multiplied the chunk kDc-1 positions further down the stream is fetched into the buffer chunk
// c-1 just released: from `cur` while it lasts, then from `nxt` starting at its chunk nxt_c0 (has_nxt == false: the
// stream ends; the spare slots re-read a valid address). lds + xs: this lane's LDS operand (row i, k offset 4g) -- an
// OFFSET into the workgroup's LDS array, not a pointer: a generic pointer turns the operand reads into flat loads,
// which count on vmcnt and drain the weight stream at every chunk.
template <int NT>
__device__ __forceinline__ void gemm_seg(WStream<NT>& ws, const WPtr<NT>& cur, int c_lo, int c_hi, const WPtr<NT>& nxt,
                                         int nxt_c0, bool has_nxt, const float* lds, int xs, int lane4, f32x4 (&acc)[NT]) {
  f32x4 a_cur = *(const f32x4*)(lds + xs + 16 * c_lo);
  for (int c0 = c_lo; c0 < c_hi; c0 += kDc) {
#pragma unroll
    for (int u = 0; u < kDc; ++u) {
      const int c = c0 + u;
      const f32x4 a_nxt = *(const f32x4*)(lds + xs + 16 * (c + 1));   // one chunk past the end at the last chunk: unused
      const int q = c + kDc - 1;
      const bool in_cur = q < c_hi;
      const bool use_nxt = !in_cur && has_nxt;
      const int qc = in_cur ? q : (has_nxt ? nxt_c0 + (q - c_hi) : c_lo);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int t = e; t < NT; t += 4) {
          const float* p = use_nxt ? nxt.p[t] : cur.p[t];
          ws.b[(u + kDc - 1) % kDc][t] = gload4(p + (size_t)qc * 256 + lane4);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws.b[u][t][e], a_cur[e], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      a_cur = a_nxt;
    }
  }
}

// narrow products whose contraction is split over the 4 waves (output layers, dL/d action): NTO output tiles,
// wave w multiplies chunks [w*CW, (w+1)*CW) and leaves its partial tiles in LDS; the caller adds the 4 partials in
// wave order. wf: packed [NTO tiles][C chunks]; xs as above. red: [4][NTO][64] float4.
template <int NTO_MAX>
struct NarrowFrags { f32x4 w[NTO_MAX][4]; };   // up to 4 chunks per wave (W <= 256)

template <int NTO_MAX>
__device__ __forceinline__ void narrow_load(NarrowFrags<NTO_MAX>& f, const float* wf, int C, int nto, int CW, int wave, int lane4) {
#pragma unroll
  for (int t = 0; t < NTO_MAX; ++t) {
    if (t < nto) {   // wave-uniform
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int cc = c < CW ? c : 0;
        f.w[t][c] = gload4(wf + ((size_t)t * C + (size_t)wave * CW + cc) * 256 + lane4);
      }
    }
  }
}
template <int NTO_MAX>
__device__ __forceinline__ void narrow_mma(const NarrowFrags<NTO_MAX>& f, int nto, int CW, int wave, float* lds, int xs, int red, int lane) {
  f32x4 a[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) a[c] = *(const f32x4*)(lds + xs + 16 * (wave * CW + (c < CW ? c : 0)));
#pragma unroll
  for (int t = 0; t < NTO_MAX; ++t) {
    if (t < nto) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c < CW) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[t][c][e], a[c][e], acc, 0, 0, 0);
        }
      }
      *(f32x4*)(lds + red + ((wave * NTO_MAX + t) * 64 + lane) * 4) = acc;
    }
  }
}
// element (row m, output n) of the product: the 4 wave partials added in wave order
template <int NTO_MAX>
__device__ __forceinline__ float narrow_get(const float* lds, int red, int m, int n) {
  const int t = n >> 4, ln = (((n & 15) >> 2) << 4) + m, e = n & 3;
  float s = lds[red + ((0 * NTO_MAX + t) * 64 + ln) * 4 + e];
  s += lds[red + ((1 * NTO_MAX + t) * 64 + ln) * 4 + e];
  s += lds[red + ((2 * NTO_MAX + t) * 64 + ln) * 4 + e];
  s += lds[red + ((3 * NTO_MAX + t) * 64 + ln) * 4 + e];
  return s;
}

// sum over the 16 lanes of a DPP row (the 16 threads that share a batch row in the row phases)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// block -> (unit, slice): the dispatcher places block b on XCD b % 8; all slices of a unit run on the same XCD(s), so
// a net's packed weights are fetched into ONE L2 (n_units <= 4: two XCDs per unit and so on). Placement is speed only.
__device__ __forceinline__ bool chain_decode(int b, int n_units, int n_slices, int& unit, int& slice) {
  const int rep = n_units >= 5 ? 1 : (n_units >= 3 ? 2 : (n_units == 2 ? 4 : 8));
  const int xcd = b & 7, rnd = b >> 3;
  if (xcd >= n_units * rep) return false;
  unit = xcd % n_units;
  slice = rnd * rep + xcd / n_units;
  return slice < n_slices;
}
inline int chain_grid(int n_units, int n_slices) {
  const int rep = n_units >= 5 ? 1 : (n_units >= 3 ? 2 : (n_units == 2 ? 4 : 8));
  return 8 * ((n_slices + rep - 1) / rep);
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_fwd
// ---------------------------------------------------------------------------------------------------------------
enum : int { SEG_FULL = 0, SEG_OBS_ONLY = 1, SEG_ACT_FROM_SAVED = 2, SEG_FULL_SAVE = 3 };
enum : int { HEAD_NONE = 0, HEAD_POLICY = 1, HEAD_Q = 2 };

struct FwdUnit {
  const float* wf[kChMaxL + 1];     // fwd-packed weights per layer (index L: output layer)
  const float* bias[kChMaxL + 1];
  const float* x;                   // input rows [B][ldx]: observation part at column 0, action part at column F
  int seg;                          // SEG_*
  int c_act;                        // chunks of this net's action segment (0: policy nets)
  float* zsave; const float* zinit; // [B][W] first-layer accumulators after the observation part
  float* H[kChMaxL]; float* G[kChMaxL];   // gelu(z), gelu'(z) per layer; nullptr: not kept
  int head;                         // HEAD_*
  float* logits; float* logp; const float* eps;   // policy head: [B][2A], [B], [B][A]
  float* xact; float* xact2;        // policy head: rows whose action columns (at F) receive the sampled action
  float* qout; float* qstd;         // q head: raw (mean, pre-softplus std) [B][2]; (std, d std/d raw) [B][2] or nullptr
  float* part_heads;                // policy head: [slices][2] sums of tanh(mu), sigma; nullptr: none
};
constexpr int kMaxFwdUnits = 6;
struct FwdArgs {
  FwdUnit u[kMaxFwdUnits];
  int n_units, n_slices;
  int B, F, A, L, ldx;
  int c_obs, c_act;                 // chunks of the first layer's observation segment / widest action segment (multiples of kDc)
  int v1_stats;
  const float* act_scale; const float* act_center; float lo_ls, hi_ls;
  long long* timeline;
};

// LDS carve-up shared by the chain kernels (floats)
struct ChainLds {
  int ld_in, ld_h, off_in, off_h0, off_h1, off_red, off_bias, total;
};
__host__ __device__ inline ChainLds chain_lds(int k_in /*floats of the widest operand row*/, int W) {
  ChainLds s;
  s.ld_in = k_in + 8; s.ld_h = W + 8;
  s.off_in = 0;
  s.off_h0 = s.off_in + kChRows * s.ld_in + 16;
  s.off_h1 = s.off_h0 + kChRows * s.ld_h + 16;
  s.off_red = s.off_h1 + kChRows * s.ld_h + 16;
  s.off_bias = s.off_red + 4 * 4 * 64 * 4 + 64;   // red: [4 waves][<=4 tiles][64][4] + row scratch
  s.total = s.off_bias + kChMaxL * W;              // hidden-layer biases (forward)
  return s;
}

template <int NT>
__global__ void __launch_bounds__(kThreads) k_chain_fwd(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit, slice;
  if (!chain_decode((int)blockIdx.x, a.n_units, a.n_slices, unit, slice)) return;
  const FwdUnit& u = a.u[unit];
  constexpr int W = 64 * NT, CH = 4 * NT;          // layer width, chunks of a hidden layer
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, lane4 = lane * 4;
  const int tile0 = NT * wave;
  const int row0 = slice * kChRows;
  const int L = a.L, F = a.F, A = a.A;
  const int C0 = a.c_obs + u.c_act;
  const ChainLds S = chain_lds(16 * (a.c_obs + a.c_act), W);
  // LDS buffers are addressed by OFFSET into lds[] (see gemm_seg)
  const int xin = S.off_in, red = S.off_red;
  // ---- the weight stream starts before anything else
  const bool do_obs = u.seg != SEG_ACT_FROM_SAVED;
  const bool do_act = u.seg != SEG_OBS_ONLY && u.c_act > 0;
  CTL(a.timeline, 0);
  WStream<NT> ws;
  const WPtr<NT> w0 = wptr<NT>(u.wf[0], C0, tile0);
  stream_prologue<NT>(ws, w0, do_obs ? 0 : a.c_obs, lane4);
  // ---- stage the input rows: xin[r][k'] (observation part, zero padding to 16*c_obs, action part, zero padding)
  {
    const int Fp = 16 * a.c_obs;
    if (do_obs) {
      const int fq = (Fp + 3) >> 2;
      for (int e = tid; e < kChRows * fq; e += kThreads) {
        const int r = e / fq, k = (e % fq) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* s = u.x + (size_t)(row0 + r) * a.ldx + k;
        if (k + 3 < F) v = *(const f32x4u*)s;
        else for (int c = 0; c < 4; ++c) if (k + c < F) v[c] = s[c];
        *(f32x4*)(lds + xin + r * S.ld_in + k) = v;
      }
    }
    if (do_act) {
      const int aq = 4 * u.c_act;   // float4 groups of the action segment
      for (int e = tid; e < kChRows * aq; e += kThreads) {
        const int r = e / aq, k = (e % aq) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const float* s = u.x + (size_t)(row0 + r) * a.ldx + F + k;
        for (int c = 0; c < 4; ++c) if (k + c < A) v[c] = s[c];
        *(f32x4*)(lds + xin + r * S.ld_in + Fp + k) = v;
      }
    }
  }
  // hidden-layer biases -> LDS: a global load issued in an epilogue would be the youngest in flight, and waiting for
  // it (loads return in order) would drain the weight stream
  for (int e = tid; e < L * (W / 4); e += kThreads) {
    const int l = e / (W / 4), k = (e % (W / 4)) * 4;
    *(f32x4*)(lds + S.off_bias + l * W + k) = gload4(u.bias[l] + k);
  }
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (u.seg == SEG_ACT_FROM_SAVED) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = gload4(u.zinit + (size_t)(row0 + i) * W + 16 * (tile0 + t) + 4 * g);
  }
  lds_barrier();
  CTL(a.timeline, 1);
  const int xs_in = xin + i * S.ld_in + 4 * g;
  // ---- first layer
  const bool more = L > 1;
  const WPtr<NT> w1 = wptr<NT>(u.wf[more ? 1 : 0], CH, tile0);
  if (do_obs) {
    // continues into the action segment (the same tensor, next chunks) or the second layer
    const bool into_act = do_act;
    gemm_seg<NT>(ws, w0, 0, a.c_obs, into_act ? w0 : w1, into_act ? a.c_obs : 0, into_act || (more && u.seg != SEG_OBS_ONLY), lds, xs_in, lane4, acc);
    if (u.zsave) {
#pragma unroll
      for (int t = 0; t < NT; ++t) *(f32x4*)(u.zsave + (size_t)(row0 + i) * W + 16 * (tile0 + t) + 4 * g) = acc[t];
    }
    CTL(a.timeline, 2);
    if (u.seg == SEG_OBS_ONLY) return;
  }
  if (do_act) gemm_seg<NT>(ws, w0, a.c_obs, C0, w1, 0, more, lds, xs_in, lane4, acc);
  CTL(a.timeline, 3);
  // ---- epilogues + hidden layers
  NarrowFrags<4> hf;
  const int nto = u.head == HEAD_POLICY ? (2 * A + 15) >> 4 : 1;
  for (int l = 0; l < L; ++l) {
    if (l == L - 1 && u.head != HEAD_NONE) narrow_load<4>(hf, u.wf[L], CH, nto, NT, wave, lane4);   // under the last epilogue
    const int hn = (l & 1) ? S.off_h1 : S.off_h0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (tile0 + t) + 4 * g;
      const f32x4 z = acc[t] + *(const f32x4*)(lds + S.off_bias + l * W + n);
      f32x4 hv, gd;
      gelu4(z, hv, gd);
      *(f32x4*)(lds + hn + i * S.ld_h + n) = hv;
      if (u.H[l]) *(f32x4*)(u.H[l] + (size_t)(row0 + i) * W + n) = hv;
      if (u.G[l]) *(f32x4*)(u.G[l] + (size_t)(row0 + i) * W + n) = gd;
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();
    CTL(a.timeline, 4 + 2 * l);
    if (l + 1 < L) {
      const WPtr<NT> wc = wptr<NT>(u.wf[l + 1], CH, tile0);
      const bool has_nxt = l + 2 < L;
      const WPtr<NT> wn = wptr<NT>(u.wf[has_nxt ? l + 2 : l + 1], CH, tile0);
      gemm_seg<NT>(ws, wc, 0, CH, wn, 0, has_nxt, lds, hn + i * S.ld_h + 4 * g, lane4, acc);
      CTL(a.timeline, 5 + 2 * l);
    }
  }
  if (u.head == HEAD_NONE) return;
  // ---- output layer: contraction split over the waves, partials through LDS
  const int hl = ((L - 1) & 1) ? S.off_h1 : S.off_h0;
  narrow_mma<4>(hf, nto, NT, wave, lds, hl + i * S.ld_h + 4 * g, red, lane);
  lds_barrier();
  CTL(a.timeline, 12);
  const int m = tid >> 4, j = tid & 15;       // row phase: 16 threads per batch row (one DPP row)
  const int r = row0 + m;
  if (u.head == HEAD_Q) {
    if (j == 0) {
      const float mean = narrow_get<4>(lds, red, m, 0) + u.bias[L][0];
      const float raw = narrow_get<4>(lds, red, m, 1) + u.bias[L][1];
      u.qout[2 * r] = mean; u.qout[2 * r + 1] = raw;
      if (u.qstd) { u.qstd[2 * r] = softplus(raw); u.qstd[2 * r + 1] = softplus_grad(raw); }
    }
    CTL(a.timeline, 13);
    return;
  }
  // policy: (mu, raw log-std) -> tanh-Gaussian rsample (act_distribution_cls.py:44-54)
  float lp = 0.f, s_tanh = 0.f, s_sig = 0.f;
  for (int d = j; d < A; d += 16) {
    const float mu = narrow_get<4>(lds, red, m, d) + u.bias[L][d];
    const float raw = narrow_get<4>(lds, red, m, A + d) + u.bias[L][A + d];
    const float eps = u.eps[(size_t)r * A + d];
    const TanhGaussFwd f = tanh_gauss_fwd(mu, raw, eps, a.act_scale[d], a.act_center[d], a.lo_ls, a.hi_ls);
    lp += f.lp;
    u.xact[(size_t)r * a.ldx + F + d] = f.a;
    if (u.xact2) u.xact2[(size_t)r * a.ldx + F + d] = f.a;
    u.logits[(size_t)r * 2 * A + d] = mu;
    u.logits[(size_t)r * 2 * A + A + d] = raw;
    if (!a.v1_stats) { s_tanh += tanhf(mu); s_sig += f.sigma; }
    else {
      if (d == 0) s_tanh += tanhf(mu);
      if (A >= 2) { if (d == 1) s_sig += mu; } else s_sig += f.sigma;
    }
  }
  lp = row16_sum(lp);
  if (j == 0) u.logp[r] = lp;
  CTL(a.timeline, 13);
  if (u.part_heads) {
    s_tanh = wave_sum(s_tanh); s_sig = wave_sum(s_sig);
    float* sc = lds + red + 4 * 4 * 64 * 4;   // row scratch behind the partial tiles
    if (lane == 0) { sc[wave] = s_tanh; sc[4 + wave] = s_sig; }
    lds_barrier();
    if (tid == 0) {
      u.part_heads[2 * slice] = sc[0] + sc[1] + sc[2] + sc[3];
      u.part_heads[2 * slice + 1] = sc[4] + sc[5] + sc[6] + sc[7];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_bwd_q: loss + dZ chains of q1(obs,act), q2(obs,act), q1(obs,new_act), q2(obs,new_act)
// ---------------------------------------------------------------------------------------------------------------
struct BwdQUnit {
  const float* wb[kChMaxL];        // bwd-packed W_l^T, l = 1..L-1
  const float* wout;               // row-major output layer [2][W] of this chain's net
  const float* G[kChMaxL];         // gelu' of this chain
  float* dZ[kChMaxL];
  float* dout;                     // [B][2]
  const float* w1at; float* dA;    // actor chains: packed (W0[:, F:])^T [16*nta x W], dL/d new_act partial [B][32]
  int which;                       // 0 q1c, 1 q2c, 2 q1p, 3 q2p
};
struct BwdQArgs {
  BwdQUnit u[4];
  int n_units, n_slices;
  int B, A, L;
  // loss inputs (dsac_v2.py:218-318), as k_loss
  const float* qout_c[2]; const float* qstd_c[2]; const float* qout_t[2]; const float* qout_p[2];
  const float* rew; const float* done; const float* logp2; const float* logp_new; const float* z5; const float* z6;
  const float* log_alpha;
  float* part_loss; float* grads_tail; const DevState* st;
  float inv_B, inv_Bg; const float* std_sums;
  int auto_alpha; float alpha_fixed, gamma, tau_b, one_minus_tau_b;
  int n_chain_blocks;
  long long* timeline;
  RideArgs ride;
};

template <int NT>
__global__ void __launch_bounds__(kThreads) k_chain_bwd_q(BwdQArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_chain_blocks) { loss_rider(a.ride); return; }
  int unit, slice;
  if (!chain_decode((int)blockIdx.x, a.n_units, a.n_slices, unit, slice)) return;
  const BwdQUnit& u = a.u[unit];
  constexpr int W = 64 * NT, CH = 4 * NT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, lane4 = lane * 4;
  const int tile0 = NT * wave;
  const int row0 = slice * kChRows;
  const int L = a.L;
  const ChainLds S = chain_lds(W, W);
  const int red = S.off_red;
  float* sc = lds + red + 4 * 4 * 64 * 4;
  // ---- weight stream: layers L-1 .. 1 (nothing to stream for a one-hidden-layer net)
  CTL(a.timeline, 0);
  WStream<NT> ws;
  if (L > 1) stream_prologue<NT>(ws, wptr<NT>(u.wb[L - 1], CH, tile0), 0, lane4);
  // ---- batch sums of std1 / std2 -> mean_std EMA (dsac_v2.py:233-241); identical in every workgroup
  float s1 = 0.f, s2 = 0.f;
  if (a.std_sums == nullptr) {
    for (int r = tid; r < a.B; r += kThreads) { s1 += a.qstd_c[0][2 * r]; s2 += a.qstd_c[1][2 * r]; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { sc[wave] = s1; sc[4 + wave] = s2; }
  }
  // this thread's share of the output-layer backward: row m, hidden units 4j + 64q
  const int m = tid >> 4, j = tid & 15;
  const int r = row0 + m;
  f32x4 w0v[NT], w1v[NT], gv[NT];
#pragma unroll
  for (int q = 0; q < NT; ++q) {
    const int k = 4 * j + 64 * q;
    w0v[q] = gload4(u.wout + k); w1v[q] = gload4(u.wout + W + k);
    gv[q] = gload4(u.G[L - 1] + (size_t)r * W + k);
  }
  const float q1 = a.qout_c[0][2 * r], q2 = a.qout_c[1][2 * r];
  const float std1 = a.qstd_c[0][2 * r], sg1 = a.qstd_c[0][2 * r + 1];
  const float std2 = a.qstd_c[1][2 * r], sg2 = a.qstd_c[1][2 * r + 1];
  const float q1n = a.qout_t[0][2 * r], raw1n = a.qout_t[0][2 * r + 1];
  const float q2n = a.qout_t[1][2 * r], raw2n = a.qout_t[1][2 * r + 1];
  const float q1p = a.qout_p[0][2 * r], q2p = a.qout_p[1][2 * r];
  const float in_z5 = a.z5[r], in_z6 = a.z6[r], rew = a.rew[r], in_done = a.done[r];
  const float lp2 = a.logp2[r], lpn = a.logp_new[r];
  const float la = a.log_alpha[0];
  const float ms1_old = a.st->ms1, ms2_old = a.st->ms2;
  const int ms_init = a.st->ms_init;
  lds_barrier();
  CTL(a.timeline, 1);
  if (a.std_sums == nullptr) { s1 = sc[0] + sc[1] + sc[2] + sc[3]; s2 = sc[4] + sc[5] + sc[6] + sc[7]; }
  else { s1 = a.std_sums[0]; s2 = a.std_sums[1]; }
  const float m1 = s1 * a.inv_Bg, m2 = s2 * a.inv_Bg;
  float ms1, ms2;
  if (!ms_init) { ms1 = m1; ms2 = m2; }
  else { ms1 = a.one_minus_tau_b * ms1_old + a.tau_b * m1; ms2 = a.one_minus_tau_b * ms2_old + a.tau_b * m2; }
  const float alpha = a.auto_alpha ? expf(la) : a.alpha_fixed;
  // ---- per-sample math (the 16 threads of a row compute it redundantly)
  const float std1n = softplus(raw1n), std2n = softplus(raw2n);
  const float qn = fminf(q1n, q2n);
  const float z5 = clampf(in_z5, -3.f, 3.f), z6 = clampf(in_z6, -3.f, 3.f);
  const float qs = (q1n < q2n) ? (q1n + z5 * std1n) : (q2n + z6 * std2n);
  const float nd = 1.0f - in_done;
  const float tq = rew + nd * a.gamma * (qn - alpha * lp2);
  const float tqs = rew + nd * a.gamma * (qs - alpha * lp2);
  const CriticTerm c1 = critic_term(q1, std1, ms1, tq, tqs);
  const CriticTerm c2 = critic_term(q2, std2, ms2, tq, tqs);
  const float wq1 = q1p < q2p ? 1.0f : (q1p > q2p ? 0.0f : 0.5f);
  float d0, d1;
  if (u.which == 0) { d0 = c1.dq * a.inv_B; d1 = c1.dstd * a.inv_B * sg1; }
  else if (u.which == 1) { d0 = c2.dq * a.inv_B; d1 = c2.dstd * a.inv_B * sg2; }
  else if (u.which == 2) { d0 = -wq1 * a.inv_B; d1 = 0.0f; }
  else { d0 = -(1.0f - wq1) * a.inv_B; d1 = 0.0f; }
  if (j == 0) {
    u.dout[2 * r] = d0; u.dout[2 * r + 1] = d1;
    if (u.which == 0) {
      float* pl = a.part_loss + (size_t)r * kLossPart;
      pl[0] = c1.loss; pl[1] = c2.loss; pl[2] = q1; pl[3] = q2; pl[4] = std1; pl[5] = std2;
      pl[6] = alpha * lpn - fminf(q1p, q2p);
      pl[7] = lpn;
      pl[8] = r == 0 ? alpha : 0.0f;
      pl[9] = 0.0f; pl[10] = std1; pl[11] = std2;
      if (r == 0) { a.grads_tail[0] = ms1; a.grads_tail[1] = ms2; }
    }
  }
  // ---- dZ of the last hidden layer: (dOut . Wout) * gelu'
  {
#pragma unroll
    for (int q = 0; q < NT; ++q) {
      const int k = 4 * j + 64 * q;
      f32x4 ov;
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (d0 * w0v[q][e] + d1 * w1v[q][e]) * gv[q][e];
      *(f32x4*)(lds + S.off_h0 + m * S.ld_h + k) = ov;
      *(f32x4*)(u.dZ[L - 1] + (size_t)r * W + k) = ov;
    }
  }
  NarrowFrags<2> af;
  const int nta = (a.A + 15) >> 4;
  if (u.w1at && L == 1) narrow_load<2>(af, u.w1at, CH, nta, NT, wave, lane4);
  lds_barrier();
  CTL(a.timeline, 2);
  // ---- hidden layers: dZ[l-1] = (dZ[l] W_l) * gelu'(z[l-1])
  f32x4 acc[NT];
  int cur = 0;
  for (int l = L - 1; l >= 1; --l) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) gq[t] = gload4(u.G[l - 1] + (size_t)(row0 + i) * W + 16 * (tile0 + t) + 4 * g);
    const bool has_nxt = l > 1;
    gemm_seg<NT>(ws, wptr<NT>(u.wb[l], CH, tile0), 0, CH, wptr<NT>(u.wb[has_nxt ? l - 1 : l], CH, tile0), 0, has_nxt,
                 lds, (cur ? S.off_h1 : S.off_h0) + i * S.ld_h + 4 * g, lane4, acc);
    CTL(a.timeline, 3 + 2 * (L - 1 - l));
    if (l == 1 && u.w1at) narrow_load<2>(af, u.w1at, CH, nta, NT, wave, lane4);
    const int hn = cur ? S.off_h0 : S.off_h1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (tile0 + t) + 4 * g;
      const f32x4 dz = acc[t] * gq[t];
      *(f32x4*)(lds + hn + i * S.ld_h + n) = dz;
      *(f32x4*)(u.dZ[l - 1] + (size_t)(row0 + i) * W + n) = dz;
    }
    cur ^= 1;
    lds_barrier();
    CTL(a.timeline, 4 + 2 * (L - 1 - l));
  }
  if (!u.w1at) return;
  // ---- dL/d new_act through this critic: dZ0 . W0[:, F:F+A]   (contraction over the hidden units, split over waves)
  narrow_mma<2>(af, nta, NT, wave, lds, (cur ? S.off_h1 : S.off_h0) + i * S.ld_h + 4 * g, red, lane);
  lds_barrier();
  for (int d = j; d < 16 * nta; d += 16) u.dA[(size_t)r * 32 + d] = d < a.A ? narrow_get<2>(lds, red, m, d) : 0.0f;
  CTL(a.timeline, 12);
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_bwd_pi: dL/d new_act -> rsample backward -> policy output-layer backward -> policy dZ chain
// blocks >= n_chain_blocks: ride-along weight-gradient tiles (the critics' dW + Adam)
// ---------------------------------------------------------------------------------------------------------------
struct BwdPiArgs {
  const float* dA[2];              // [B][32] from k_chain_bwd_q (q1p, q2p)
  const float* logits_pi; const float* eps_new; const float* log_alpha;
  const float* woutT; int CoT;     // packed Wout_pi^T [W x 16*CoT]
  const float* wb[kChMaxL];        // bwd-packed policy layers
  const float* G[kChMaxL];
  float* dZ[kChMaxL];
  float* dout_pi; float* d_new_act;
  int n_slices, B, A, L;
  float inv_B; int auto_alpha; float alpha_fixed;
  const float* act_scale; float lo_ls, hi_ls;
  const float* part_loss; int n_part; float target_entropy; float* grad_log_alpha;
  int n_chain_blocks;
  const GemmProb* extra; int n_extra;
  long long* timeline;
  FusedOpt fo;
};

template <int NT>
__global__ void __launch_bounds__(kThreads) k_chain_bwd_pi(BwdPiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_chain_blocks) {
    const GemmProb gp = a.extra[blockIdx.x - a.n_chain_blocks];
    run_tile<true, true, EPI_STORE>(gp, gp.tiles_n, gp.tile_end, lds, nullptr, 0, &a.fo);
    return;
  }
  const int slice = (int)blockIdx.x;
  if (slice >= a.n_slices) return;
  constexpr int W = 64 * NT, CH = 4 * NT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, lane4 = lane * 4;
  const int tile0 = NT * wave;
  const int row0 = slice * kChRows;
  const int L = a.L, A = a.A;
  const ChainLds S = chain_lds(16 * a.CoT, W);
  float* xdo = lds + S.off_in;
  CTL(a.timeline, 0);
  WStream<NT> ws;
  const WPtr<NT> wo = wptr<NT>(a.woutT, a.CoT, tile0);
  stream_prologue<NT>(ws, wo, 0, lane4);
  // alpha gradient (dsac_v2.py:312-318): -mean(logp_new + target_entropy)
  if (slice == 0 && wave == 0) {
    float s = 0.f;
    for (int r0 = 0; r0 < a.n_part; r0 += 64) {
      const int rr = r0 + lane;
      s += rr < a.n_part ? a.part_loss[(size_t)rr * kLossPart + 7] : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) a.grad_log_alpha[0] = a.auto_alpha ? -(s * a.inv_B + a.target_entropy) : 0.0f;
  }
  const int m = tid >> 4, j = tid & 15;
  const int r = row0 + m;
  const float alpha = a.auto_alpha ? expf(a.log_alpha[0]) : a.alpha_fixed;
  // zero the operand rows' padding, then fill (dmu | draw)
  for (int e = tid; e < kChRows * 16 * a.CoT; e += kThreads) xdo[(e / (16 * a.CoT)) * S.ld_in + e % (16 * a.CoT)] = 0.0f;
  lds_barrier();
  for (int d = j; d < A; d += 16) {
    const float dA = a.dA[0][(size_t)r * 32 + d] + a.dA[1][(size_t)r * 32 + d];
    const float mu = a.logits_pi[(size_t)r * 2 * A + d], raw = a.logits_pi[(size_t)r * 2 * A + A + d];
    float dmu, draw;
    tanh_gauss_bwd(mu, raw, a.eps_new[(size_t)r * A + d], a.act_scale[d], a.lo_ls, a.hi_ls, dA, alpha * a.inv_B, dmu, draw);
    a.dout_pi[(size_t)r * 2 * A + d] = dmu;
    a.dout_pi[(size_t)r * 2 * A + A + d] = draw;
    a.d_new_act[(size_t)r * A + d] = dA;
    xdo[m * S.ld_in + d] = dmu;
    xdo[m * S.ld_in + A + d] = draw;
  }
  lds_barrier();
  CTL(a.timeline, 1);
  f32x4 acc[NT];
  // ---- policy output layer backward: (dmu | draw) . Wout, then * gelu'(z_last)
  {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) gq[t] = gload4(a.G[L - 1] + (size_t)(row0 + i) * W + 16 * (tile0 + t) + 4 * g);
    const bool has_nxt = L > 1;
    gemm_seg<NT>(ws, wo, 0, a.CoT, wptr<NT>(has_nxt ? a.wb[L - 1] : a.woutT, has_nxt ? CH : a.CoT, tile0), 0, has_nxt,
                 lds, S.off_in + i * S.ld_in + 4 * g, lane4, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (tile0 + t) + 4 * g;
      const f32x4 dz = acc[t] * gq[t];
      *(f32x4*)(lds + S.off_h0 + i * S.ld_h + n) = dz;
      *(f32x4*)(a.dZ[L - 1] + (size_t)(row0 + i) * W + n) = dz;
    }
    lds_barrier();
    CTL(a.timeline, 2);
  }
  int cur = 0;
  for (int l = L - 1; l >= 1; --l) {
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 gq[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) gq[t] = gload4(a.G[l - 1] + (size_t)(row0 + i) * W + 16 * (tile0 + t) + 4 * g);
    const bool has_nxt = l > 1;
    gemm_seg<NT>(ws, wptr<NT>(a.wb[l], CH, tile0), 0, CH, wptr<NT>(a.wb[has_nxt ? l - 1 : l], CH, tile0), 0, has_nxt,
                 lds, (cur ? S.off_h1 : S.off_h0) + i * S.ld_h + 4 * g, lane4, acc);
    const int hn = cur ? S.off_h0 : S.off_h1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = 16 * (tile0 + t) + 4 * g;
      const f32x4 dz = acc[t] * gq[t];
      *(f32x4*)(lds + hn + i * S.ld_h + n) = dz;
      *(f32x4*)(a.dZ[l - 1] + (size_t)(row0 + i) * W + n) = dz;
    }
    cur ^= 1;
    if (l > 1) lds_barrier();
    CTL(a.timeline, 3 + (L - 1 - l));
  }
}

}  // namespace dsact
