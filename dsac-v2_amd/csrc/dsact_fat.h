// dsact_fat.h -- the row-slice chains of dsact_chain.h re-cut for the THROUGHPUT regime (batch >= 1024).
//
// At batch 256 a chain workgroup owns 4-8 rows and streams its net's whole weight set (v_mfma_f32_4x4x1: lane = output
// feature, one LDS broadcast read + one 1 KB weight load per 8 MFMAs): matrix pipe, LDS pipe and the CU's 64 B/clk L2
// port are all at par -- right when every slice needs a CU of its own, wrong when there are thousands of rows (batch
// 4096: 512 slices x 6 units each re-stream 0.9 MB; measured 28 % of the fp32 peak).
// Here a workgroup owns R = 16 or 32 rows and multiplies with v_mfma_f32_16x16x4_f32 (same fp32 rate, 4x the operand
// reuse): wave w owns output features [64w, 64w+64) as 4 tiles of 16, per 16-k chunk it issues RT*16 MFMAs (512 /
// 1024 cycles) for 4 KB of weights (L2 -> registers, style-16 fragment-major packs, two chunks in flight) and RT
// ds_read_b128 -- LDS ~6 % busy, L2 port ~25 %, the matrix pipe is the bound. The activations of a slice live in ONE
// LDS buffer that is overwritten in place (the whole layer output sits in the accumulators when the last read of its
// input is done), the first layer reads its rows straight from the staged minibatch (L2), so a workgroup needs 33 KB of
// LDS and three fit on a CU: another workgroup's MFMAs cover this one's epilogue and barriers.
// Same arguments, same packs' element format ([feature][batch] 16-byte elements), same math as the batch-256 kernels:
// FwdArgs / BwdQArgs / BwdPiArgs are shared; in this mode s_obs / s_act / SoT count 16-k chunks, and every weight
// pack is style 16 (MirrorDesc::fwd_44 / bwd_44 == 0).
// Reference math: networks/mlp.py:79-127, utils/act_distribution_cls.py:44-54, dsac_v2.py:150-318.
#pragma once
#include "dsact_chain.h"

namespace dsact {

struct FatLds { int ldh, ldx, ldx0, off_h, off_x, off_sc, total; };
// kx: floats of the staged narrow operand rows (policy backward: the (dmu | draw) rows), 0: none
// k0: floats of the forward's staged INPUT rows (16 per first-layer chunk; round 6), 0: the first layer reads its rows from
//     global memory. The input rows share the activation buffer (the first layer's output sits in the accumulators when
//     the last read of its input is done, like every later layer's), so the buffer is sized by the wider of the two
__host__ __device__ inline FatLds fat_lds(int W, int R, int kx, int k0 = 0) {
  FatLds s;
  s.ldh = W + 4; s.ldx = kx + 4; s.ldx0 = k0 ? k0 + 4 : 0;
  s.off_h = 0;
  s.off_x = s.off_h + R * (s.ldx0 > s.ldh ? s.ldx0 : s.ldh);
  s.off_sc = s.off_x + (kx ? R * s.ldx : 0);
  s.total = s.off_sc + 16 + 2 * R;   // [16] wave partial sums, [2R] per-row (d0, d1)
  return s;
}

// one 16-k chunk: acc[rt][nt] (rows 16rt + 4g + reg, feature 16nt + i) += A[16rt + i][k] * W[16nt + i][k], k = 4g + e
template <int RT>
__device__ __forceinline__ void fat_mma(f32x4 (&acc)[RT][4], const f32x4 (&a)[RT], const f32x4 (&b)[4]) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt][e], b[nt][e], acc[rt][nt], 0, 0, 0);
}

// wt: style-16 pack of the wave's first 16-feature tile (tile T of the tensor sits T * C * 256 floats in)
__device__ __forceinline__ void fat_load_b(f32x4 (&b)[4], const float* wt, int C, int c, int lane4) {
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b[nt] = gload4(wt + ((size_t)nt * C + c) * 256 + lane4);
}

// acc += A (LDS rows, row-major, leading dimension ld; xs = offset of element [i][4g]) x W over chunks [c_lo, c_hi).
// Two chunks of weights are in flight; a slot is refilled right after the MFMAs that used it (past the end: the last
// chunk again -- a valid address, no branch around the load).
template <int RT>
__device__ __forceinline__ void fat_gemm_lds(f32x4 (&acc)[RT][4], const float* wt, int C, int c_lo, int c_hi, const float* lds,
                                             int xs, int ld, int lane4) {
  if (c_hi <= c_lo) return;
  const int last = c_hi - 1;
  f32x4 b0[4], b1[4];
  fat_load_b(b0, wt, C, c_lo, lane4);
  fat_load_b(b1, wt, C, c_lo + 1 < c_hi ? c_lo + 1 : last, lane4);
  for (int c = c_lo; c < c_hi; c += 2) {
    f32x4 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = *(const f32x4*)(lds + xs + 16 * rt * ld + 16 * c);
    fat_mma<RT>(acc, a, b0);
    fat_load_b(b0, wt, C, c + 2 < c_hi ? c + 2 : last, lane4);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < c_hi) {   // wave-uniform
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) a[rt] = *(const f32x4*)(lds + xs + 16 * rt * ld + 16 * (c + 1));
      fat_mma<RT>(acc, a, b1);
      fat_load_b(b1, wt, C, c + 3 < c_hi ? c + 3 : last, lane4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ... with a run-time chunk range and the first two chunks' weights requested by the caller (the first layer of a staged slice:
// the requests go out before the workgroup waits for its input rows). Same MFMA sequence as fat_gemm_lds; the loop body is
// branch-free (whole pairs, the odd chunk behind the loop): with the second half under a wave-uniform `if` hipcc's wait-count
// pass merges the loop's entry and back edges conservatively and drains the queue (vmcnt(0)) in front of every pair -- one
// exposed L2 round trip per pair, 0.38 us per chunk against the unrolled hidden layers' 0.275 (scripts/isa_wait_audit.py).
template <int RT>
__device__ __forceinline__ void fat_gemm_lds_pl(f32x4 (&acc)[RT][4], f32x4 (&b0)[4], f32x4 (&b1)[4], const float* wt, int C, int c_lo, int c_hi,
                                                const float* lds, int xs, int ld, int lane4) {
  if (c_hi <= c_lo) return;
  const int last = c_hi - 1;
  auto pair = [&](int c) {
    f32x4 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = *(const f32x4*)(lds + xs + 16 * rt * ld + 16 * c);
    fat_mma<RT>(acc, a, b0);
    fat_load_b(b0, wt, C, c + 2 < c_hi ? c + 2 : last, lane4);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = *(const f32x4*)(lds + xs + 16 * rt * ld + 16 * (c + 1));
    fat_mma<RT>(acc, a, b1);
    fat_load_b(b1, wt, C, c + 3 < c_hi ? c + 3 : last, lane4);
    __builtin_amdgcn_sched_barrier(0);
  };
  int c = c_lo;
  if (c + 1 < c_hi) {   // first trip peeled, the loop INSIDE its branch: the loop is entered only with the queue its back edge has (dw2_tile's form)
    pair(c);
    for (c += 2; c + 1 < c_hi; c += 2) pair(c);
  }
  if (c < c_hi) {   // odd chunk count: the last one (its weights sit in b0)
    f32x4 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = *(const f32x4*)(lds + xs + 16 * rt * ld + 16 * c);
    fat_mma<RT>(acc, a, b0);
  }
}

// First layer: the A operand comes straight from the staged minibatch rows [obs (F) | action (A) | 0 pad] in global
// memory (L2): chunk c < c_obs covers columns 16c .. 16c+15 of the observation, chunk c_obs + c' columns F + 16c' .. of
// the action. Column groups past the segment's end are zeroed AFTER the load has landed (the pack holds zero weights
// there, but what the row holds next to its observation is another workgroup's business: 0 * NaN would be NaN).
struct FatX {
  const float* p;       // &x[(row0 + i) * ldx + 4g]
  size_t rt_stride;     // 16 * ldx
  int c_obs, F, A, g4;  // g4 = 4 * (lane >> 4)
};
__device__ __forceinline__ int fat_x_col(const FatX& x, int c, bool& valid) {
  const bool obs = c < x.c_obs;
  const int col = obs ? 16 * c : x.F + 16 * (c - x.c_obs);
  const int lim = obs ? x.F : x.F + x.A;
  valid = col + x.g4 < lim;
  return valid ? col : 0;
}
template <int RT>
__device__ __forceinline__ void fat_load_x(f32x4 (&a)[RT], const FatX& x, int c, bool& valid) {
  const int col = fat_x_col(x, c, valid);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) a[rt] = gload4(x.p + rt * x.rt_stride + col);
}
template <int RT>
__device__ __forceinline__ void fat_gemm_x(f32x4 (&acc)[RT][4], const float* wt, int C, int c_lo, int c_hi, const FatX& x, int lane4) {
  if (c_hi <= c_lo) return;
  const int last = c_hi - 1;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 b0[4], b1[4], r0[RT], r1[RT];
  bool v0, v1;
  fat_load_x<RT>(r0, x, c_lo, v0);
  fat_load_b(b0, wt, C, c_lo, lane4);
  fat_load_b(b1, wt, C, c_lo + 1 < c_hi ? c_lo + 1 : last, lane4);
  for (int c = c_lo; c < c_hi; c += 2) {
    fat_load_x<RT>(r1, x, c + 1 < c_hi ? c + 1 : last, v1);
    f32x4 a[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = v0 ? r0[rt] : zero;
    fat_mma<RT>(acc, a, b0);
    fat_load_b(b0, wt, C, c + 2 < c_hi ? c + 2 : last, lane4);
    __builtin_amdgcn_sched_barrier(0);
    if (c + 1 < c_hi) {   // wave-uniform
      fat_load_x<RT>(r0, x, c + 2 < c_hi ? c + 2 : last, v0);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) a[rt] = v1 ? r1[rt] : zero;
      fat_mma<RT>(acc, a, b1);
      fat_load_b(b1, wt, C, c + 3 < c_hi ? c + 3 : last, lane4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// narrow products (output layers, dL/d action) for R = 16 * RT rows: the contraction over the W hidden units is split
// over the NW waves (4 chunks of 16 each); the partial tiles meet in LDS -- in the activation buffer itself, once every
// wave has read its operand rows (the caller's barriers) -- and are added in wave order.
template <int NTO_MAX, int RT>
__device__ __forceinline__ void fat_narrow_mma(const NarrowFrags<NTO_MAX>& f, int nto, int wave, const float* lds, int xs, int ld,
                                               f32x4 (&out)[RT][NTO_MAX]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    f32x4 a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = *(const f32x4*)(lds + xs + 16 * rt * ld + 16 * (wave * 4 + c));
#pragma unroll
    for (int t = 0; t < NTO_MAX; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (t < nto) {   // wave-uniform
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[t][c][e], a[c][e], acc, 0, 0, 0);
      }
      out[rt][t] = acc;
    }
  }
}
template <int NTO_MAX, int RT, int NW>
__device__ __forceinline__ void fat_narrow_store(const f32x4 (&out)[RT][NTO_MAX], int nto, int wave, float* lds, int red, int lane) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int t = 0; t < NTO_MAX; ++t)
      if (t < nto) *(f32x4*)(lds + red + (((rt * NW + wave) * NTO_MAX + t) * 64 + lane) * 4) = out[rt][t];
}
// element (row m of the slice, output n)
template <int NTO_MAX, int NW>
__device__ __forceinline__ float fat_narrow_get(const float* lds, int red, int m, int n) {
  const int rt = m >> 4, t = n >> 4, ln = (((n & 15) >> 2) << 4) + (m & 15), e = n & 3;
  float s = lds[red + (((rt * NW + 0) * NTO_MAX + t) * 64 + ln) * 4 + e];
#pragma unroll
  for (int w = 1; w < NW; ++w) s += lds[red + (((rt * NW + w) * NTO_MAX + t) * 64 + ln) * 4 + e];
  return s;
}

template <int TPR>
__device__ __forceinline__ float fat_row_sum(float v) {
  if (TPR >= 2) v += dpp_mov<0xB1>(v);
  if (TPR >= 4) v += dpp_mov<0x4E>(v);
  if (TPR >= 8) v += dpp_mov<0x141>(v);
  if (TPR >= 16) v += dpp_mov<0x140>(v);
  if (TPR >= 32) { float a, b; swap16(v, a, b); v = a + b; }
  if (TPR >= 64) { float a, b; swap32(v, a, b); v = a + b; }
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
template <int NW, int RT, bool GA = false>
__device__ __forceinline__ void fat_fwd_body(const FwdArgs& a, int unit, int slice, float* lds) {
  const FwdUnit& u = a.u[unit];
  constexpr int W = 64 * NW, R = 16 * RT, NTHR = 64 * NW, TPR = NTHR / R, CW = W / 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane4 = lane * 4, i = lane & 15, g = lane >> 4;
  const int row0 = slice * R;
  const int L = a.L, F = a.F, A = a.A;
  const bool stage = a.x0_lds != 0;   // round 6: the slice's input rows go through LDS once (the host decides: they must fit)
  const FatLds S = fat_lds(W, R, 0, stage ? 16 * (a.s_obs + a.s_act) : 0);
  const int c_obs = a.s_obs, c_act = u.s_act, C0 = c_obs + c_act;
  const bool do_obs = u.seg != SEG_ACT_FROM_SAVED;
  const bool do_act = u.seg != SEG_OBS_ONLY && c_act > 0;
  CTL(a.timeline, 0);   // (instrumented builds: scripts/gpu_r5_timeline_fat.sh)
  CTLR(a.timeline, 14);
  CTLV(a.timeline, 11, 1 + unit);
  CTLV(a.timeline, 10, ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));   // (XCC_ID, HW_ID)
  int nf[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) nf[nt] = 64 * wave + 16 * nt + i;   // this lane's output features
  // operands of the head's row phase: fetched now (a global load issued where it is used sits on the tail's critical path)
  const int mr = tid / TPR, jr = tid % TPR;
  constexpr int NQ = (32 + TPR - 1) / TPR;      // act_dim <= 32
  float pre_eps[NQ], pre_bmu[NQ], pre_braw[NQ], pre_s[NQ], pre_c[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { pre_eps[q] = 0.f; pre_bmu[q] = 0.f; pre_braw[q] = 0.f; pre_s[q] = 1.f; pre_c[q] = 0.f; }
  if (u.head == HEAD_POLICY) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int d = jr + q * TPR;
      if (d < A) {
        pre_eps[q] = u.eps[(size_t)(row0 + mr) * A + d];
        pre_bmu[q] = u.bias[L][d]; pre_braw[q] = u.bias[L][A + d];
        pre_s[q] = a.act_scale[d]; pre_c[q] = a.act_center[d];
      }
    }
  } else if (u.head == HEAD_Q && jr == 0) {
    pre_bmu[0] = u.bias[L][0]; pre_braw[0] = u.bias[L][1];
  }
  f32x4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (u.seg == SEG_ACT_FROM_SAVED) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][nt][r] = u.zinit[(size_t)(row0 + 16 * rt + 4 * g + r) * W + nf[nt]];
      }
    }
  float bl[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bl[nt] = u.bias[0][nf[nt]];
  const float* w0 = u.wf[0] + (size_t)(4 * wave) * C0 * 256;
  // ---- staged input rows (round 6). Reading them from global memory chunk by chunk made every wave of the workgroup fetch the
  // same 16 x 64 B row pieces one chunk ahead of its MFMAs (four times the bytes, half-used lines, one L2 round trip of cover
  // with one wave per SIMD: 68-83 cycles per MFMA against the hidden layers' 41, profiles/r05_timeline_fat_chain_fwd_a.txt).
  // Now the workgroup copies its rows once, coalesced, in the chunk layout the MFMAs read -- [R][16 C0], quads outside a segment
  // zero (what fat_gemm_x selects per lane) -- and the first layer runs the hidden layers' loop. Same operands, same MFMA order.
  f32x4 pb0[4], pb1[4];
  const int c_first = do_obs ? 0 : c_obs, c_end1 = do_obs ? c_obs : C0;
  if (stage) {
    fat_load_b(pb0, w0, C0, c_first, lane4);
    fat_load_b(pb1, w0, C0, c_first + 1 < c_end1 ? c_first + 1 : c_end1 - 1, lane4);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const int q_lo = do_obs ? 0 : 4 * c_obs, q_hi = do_act ? 4 * C0 : 4 * c_obs;
    const int nq = q_hi - q_lo, total = R * nq;
    for (int e0 = tid; e0 < total; e0 += 4 * NTHR) {
      f32x4 v[4];
      bool ok[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q * NTHR;
        const bool in = e < total;
        const int r = in ? e / nq : 0, qq = q_lo + (in ? e % nq : 0);
        const int c = qq >> 2, g4 = (qq & 3) * 4;
        const bool obs = c < c_obs;
        const int col = obs ? 16 * c : F + 16 * (c - c_obs);
        ok[q] = col + g4 < (obs ? F : F + A);
        v[q] = gload4(u.x + (size_t)(row0 + r) * a.ldx + (ok[q] ? col + g4 : 0));
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + q * NTHR;
        if (e < total) {
          const int r = e / nq, qq = q_lo + e % nq;
          *(f32x4*)(lds + S.off_h + r * S.ldx0 + 4 * qq) = ok[q] ? v[q] : zero;
        }
      }
    }
    lds_barrier();
  }
  // the dW tiles of every first layer read the minibatch as [input feature][batch]: one unit transposes its rows
  if (u.x0t) {
    const int K0 = F + (do_act ? A : 0);
    for (int e = tid; e < K0 * (R / 4); e += NTHR) {
      const int k = e % K0, q4 = e / K0;
      f32x4 v;
      if (stage) {
        const int kk = k < F ? k : 16 * c_obs + (k - F);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = lds[S.off_h + (4 * q4 + r) * S.ldx0 + kk];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = u.x[(size_t)(row0 + 4 * q4 + r) * a.ldx + k];
      }
      nt_store4(u.x0t + pk_index(k, row0 + 4 * q4, a.Cb), v);
    }
  }
  CTL(a.timeline, 1);
  // ---- first layer
  FatX X;
  X.p = u.x + (size_t)(row0 + i) * a.ldx + 4 * g; X.rt_stride = (size_t)16 * a.ldx;
  X.c_obs = c_obs; X.F = F; X.A = A; X.g4 = 4 * g;
  const int xs_x = S.off_h + i * S.ldx0 + 4 * g;
  if (do_obs) {
    if (stage) fat_gemm_lds_pl<RT>(acc, pb0, pb1, w0, C0, 0, c_obs, lds, xs_x, S.ldx0, lane4);
    else fat_gemm_x<RT>(acc, w0, C0, 0, c_obs, X, lane4);
    if (u.zsave) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) u.zsave[(size_t)(row0 + 16 * rt + 4 * g + r) * W + nf[nt]] = acc[rt][nt][r];
    }
    CTL(a.timeline, 2);
    if (u.seg == SEG_OBS_ONLY) { CTLR(a.timeline, 15); return; }
  }
  if (do_act) {
    if (stage && !do_obs) fat_gemm_lds_pl<RT>(acc, pb0, pb1, w0, C0, c_obs, C0, lds, xs_x, S.ldx0, lane4);
    else if (stage) fat_gemm_lds<RT>(acc, w0, C0, c_obs, C0, lds, xs_x, S.ldx0, lane4);
    else fat_gemm_x<RT>(acc, w0, C0, c_obs, C0, X, lane4);
  }
  CTL(a.timeline, 3);
  // ---- epilogues + hidden layers (the slice's activations live in ONE LDS buffer, overwritten in place)
  NarrowFrags<4> hf;
  const int nto = u.head == HEAD_POLICY ? (2 * A + 15) >> 4 : 1;
  const int xs_h = S.off_h + i * S.ldh + 4 * g;
  for (int l = 0; l < L; ++l) {
    if (l == L - 1 && u.head != HEAD_NONE) narrow_load<4>(hf, u.wf[L], CW, nto, wave, lane4);   // under the last epilogue
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 z = acc[rt][nt] + bl[nt];
        f32x4 hv, gd;
        if (GA) act4(u.act, z, hv, gd); else gelu4(z, hv, gd);
        if (u.H[l]) nt_store4(u.H[l] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb), hv);
        if (u.G[l]) nt_store4(u.G[l] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb), gd);
        acc[rt][nt] = hv;
      }
    if (l > 0 || stage) lds_barrier();      // every wave has read the last of this layer's input (l == 0: the staged rows)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[S.off_h + (16 * rt + 4 * g + r) * S.ldh + nf[nt]] = acc[rt][nt][r];
    lds_barrier();
    CTL(a.timeline, 4 + 2 * l);
    if (l + 1 < L) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) bl[nt] = u.bias[l + 1][nf[nt]];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      fat_gemm_lds<RT>(acc, u.wf[l + 1] + (size_t)(4 * wave) * CW * 256, CW, 0, CW, lds, xs_h, S.ldh, lane4);
      CTL(a.timeline, 5 + 2 * l);
    }
  }
  if (u.head == HEAD_NONE) return;
  // ---- output layer: partial tiles per wave, added in wave order
  {
    f32x4 part[RT][4];
    fat_narrow_mma<4, RT>(hf, nto, wave, lds, xs_h, S.ldh, part);
    lds_barrier();                 // everybody has read the activations: their buffer now takes the partial tiles
    fat_narrow_store<4, RT, NW>(part, nto, wave, lds, S.off_h, lane);
    lds_barrier();
  }
  CTL(a.timeline, 12);
  const int red = S.off_h;
  const int m = mr, j = jr;                    // row phase: TPR consecutive lanes per batch row
  const int r = row0 + m;
  if (u.head == HEAD_Q) {
    if (j == 0) {
      const float mean = fat_narrow_get<4, NW>(lds, red, m, 0) + pre_bmu[0];
      const float raw = fat_narrow_get<4, NW>(lds, red, m, 1) + pre_braw[0];
      u.qout[2 * r] = mean; u.qout[2 * r + 1] = raw;
      if (u.qstd) { u.qstd[2 * r] = softplus(raw); u.qstd[2 * r + 1] = softplus_grad(raw); }
    }
    CTL(a.timeline, 13);
    CTLR(a.timeline, 15);
    return;
  }
  // policy: (mu, raw log-std) -> tanh-Gaussian rsample (act_distribution_cls.py:44-54)
  float lp = 0.f, s_tanh = 0.f, s_sig = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int d = j + q * TPR;
    if (d >= A) break;
    const float mu = fat_narrow_get<4, NW>(lds, red, m, d) + pre_bmu[q];
    const float raw = fat_narrow_get<4, NW>(lds, red, m, A + d) + pre_braw[q];
    const TanhGaussFwd f = tanh_gauss_fwd(mu, raw, pre_eps[q], pre_s[q], pre_c[q], a.lo_ls, a.hi_ls);
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
  lp = fat_row_sum<TPR>(lp);
  if (j == 0) u.logp[r] = lp;
  if (u.part_heads) {
    s_tanh = wave_sum(s_tanh); s_sig = wave_sum(s_sig);
    float* sc = lds + S.off_sc;
    if (lane == 0) { sc[wave] = s_tanh; sc[4 + wave] = s_sig; }
    lds_barrier();
    if (tid == 0) {
      float t0 = 0.f, t1 = 0.f;
      for (int w = 0; w < NW; ++w) { t0 += sc[w]; t1 += sc[4 + w]; }
      u.part_heads[2 * slice] = t0;
      u.part_heads[2 * slice + 1] = t1;
    }
  }
  CTL(a.timeline, 13);
  CTLR(a.timeline, 15);
}

// blocks are unit-major (block = unit * n_slices + slice): the chip works on one or two units at a time, whose weights
// (<= 1 MB each) stay hot in every XCD's L2 while their slices stream through
template <int NW, int RT, bool GA = false>
__global__ void __launch_bounds__(64 * NW, 2) k_fat_fwd(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ns = a.u[0].n_slices;
  const int unit = (int)blockIdx.x / ns, slice = (int)blockIdx.x - unit * ns;
  if (unit >= a.n_units) return;
  fat_fwd_body<NW, RT, GA>(a, unit, slice, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// critics' backward: loss + dZ chains of q1(obs,act), q2(obs,act), q1(obs,new_act), q2(obs,new_act)
// ---------------------------------------------------------------------------------------------------------------
template <int NW, int RT>
__device__ __forceinline__ void fat_bwd_q_body(const BwdQArgs& a, int unit, int slice, float* lds) {
  constexpr int W = 64 * NW, R = 16 * RT, NTHR = 64 * NW, TPR = NTHR / R, CW = W / 16;
  const int tid = threadIdx.x;
  if (tid >= NTHR) return;                        // narrow nets: the launch is 256 wide for the riders
  const BwdQUnit& u = a.u[unit];
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane4 = lane * 4, i = lane & 15, g = lane >> 4;
  const int row0 = slice * R;
  const int L = a.L;
  const FatLds S = fat_lds(W, R, 0);
  float* sc = lds + S.off_sc;
  int nf[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) nf[nt] = 64 * wave + 16 * nt + i;
  // ---- batch sums of std1 / std2 -> mean_std EMA (dsac_v2.py:233-241); identical in every workgroup
  float s1 = 0.f, s2 = 0.f;
  if (a.std_sums == nullptr) {
    std_column_sums<NTHR>(a.qstd_c[0], a.qstd_c[1], a.B, tid, s1, s2);
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { sc[wave] = s1; sc[4 + wave] = s2; }
  }
  // row phase: TPR consecutive lanes per batch row
  const int m = tid / TPR, j = tid % TPR;
  const int r = row0 + m;
  float wo0[4], wo1[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { wo0[nt] = u.wout[nf[nt]]; wo1[nt] = u.wout[W + nf[nt]]; }
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
  if (a.std_sums == nullptr) {
    s1 = 0.f; s2 = 0.f;
    for (int w = 0; w < NW; ++w) { s1 += sc[w]; s2 += sc[4 + w]; }
  } else { s1 = a.std_sums[0]; s2 = a.std_sums[1]; }
  const float m1 = s1 * a.inv_Bg, m2 = s2 * a.inv_Bg;
  float ms1, ms2;
  if (!ms_init) { ms1 = m1; ms2 = m2; }
  else { ms1 = a.one_minus_tau_b * ms1_old + a.tau_b * m1; ms2 = a.one_minus_tau_b * ms2_old + a.tau_b * m2; }
  const float alpha = a.auto_alpha ? expf(la) : a.alpha_fixed;
  // ---- per-sample math (the TPR threads of a row compute it redundantly)
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
    sc[16 + 2 * m] = d0; sc[16 + 2 * m + 1] = d1;
    if (u.doutT) { u.doutT[pk_index(0, r, a.Cb)] = d0; u.doutT[pk_index(1, r, a.Cb)] = d1; }
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
  lds_barrier();
  // ---- dZ of the last hidden layer: (dOut . Wout) * gelu'
  f32x4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 gl = gload4(u.G[L - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb));
      f32x4 ov;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = 16 * rt + 4 * g + rr;
        ov[rr] = (sc[16 + 2 * mm] * wo0[nt] + sc[16 + 2 * mm + 1] * wo1[nt]) * gl[rr];
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) lds[S.off_h + (16 * rt + 4 * g + rr) * S.ldh + nf[nt]] = ov[rr];
      nt_store4(u.dZ[L - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb), ov);
    }
  NarrowFrags<2> af;
  const int nta = (a.A + 15) >> 4;
  if (u.w1at && L == 1) narrow_load<2>(af, u.w1at, CW, nta, wave, lane4);
  lds_barrier();
  // ---- hidden layers: dZ[l-1] = (dZ[l] W_l) * gelu'(z[l-1])
  const int xs_h = S.off_h + i * S.ldh + 4 * g;
  for (int l = L - 1; l >= 1; --l) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    fat_gemm_lds<RT>(acc, u.wb[l] + (size_t)(4 * wave) * CW * 256, CW, 0, CW, lds, xs_h, S.ldh, lane4);
    if (l == 1 && u.w1at) narrow_load<2>(af, u.w1at, CW, nta, wave, lane4);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 gq = gload4(u.G[l - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb));
        acc[rt][nt] = acc[rt][nt] * gq;
        nt_store4(u.dZ[l - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb), acc[rt][nt]);
      }
    lds_barrier();      // every wave has read the last of dZ[l]
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) lds[S.off_h + (16 * rt + 4 * g + rr) * S.ldh + nf[nt]] = acc[rt][nt][rr];
    lds_barrier();
  }
  if (!u.w1at) return;
  // ---- dL/d new_act through this critic: dZ0 . W0[:, F:F+A]   (contraction over the hidden units, split over waves)
  {
    f32x4 part[RT][2];
    fat_narrow_mma<2, RT>(af, nta, wave, lds, xs_h, S.ldh, part);
    lds_barrier();
    fat_narrow_store<2, RT, NW>(part, nta, wave, lds, S.off_h, lane);
    lds_barrier();
  }
  for (int d = j; d < 16 * nta; d += TPR) u.dA[(size_t)r * 32 + d] = d < a.A ? fat_narrow_get<2, NW>(lds, S.off_h, m, d) : 0.0f;
}

template <int NW, int RT>
__global__ void __launch_bounds__(256, 2) k_fat_bwd_q(BwdQArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_chain_blocks) { loss_rider(a.ride); return; }   // riders are 256-thread blocks
  const int unit = (int)blockIdx.x / a.n_slices, slice = (int)blockIdx.x - unit * a.n_slices;
  fat_bwd_q_body<NW, RT>(a, unit, slice, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// policy backward: dL/d new_act -> rsample backward -> policy output-layer backward -> policy dZ chain
// blocks >= n_chain_blocks: ride-along weight-gradient tiles, 256 threads each
// ---------------------------------------------------------------------------------------------------------------
template <int NW, int RT>
__device__ __forceinline__ void fat_bwd_pi_body(const BwdPiArgs& a, int slice, float* lds) {
  constexpr int W = 64 * NW, R = 16 * RT, NTHR = 64 * NW, TPR = NTHR / R, CW = W / 16;
  const int tid = threadIdx.x;
  if (slice >= a.n_slices || tid >= NTHR) return;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane4 = lane * 4, i = lane & 15, g = lane >> 4;
  const int row0 = slice * R;
  const int L = a.L, A = a.A;
  const int c_out = a.SoT;                       // 16-k chunks of the (dmu | draw) rows
  const FatLds S = fat_lds(W, R, 16 * c_out);
  float* xdo = lds + S.off_x;
  int nf[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) nf[nt] = 64 * wave + 16 * nt + i;
  const int m = tid / TPR, j = tid % TPR;
  const int r = row0 + m;
  constexpr int NQ = (32 + TPR - 1) / TPR;      // act_dim <= 32
  float pdA[NQ], pmu[NQ], praw[NQ], peps[NQ], psc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int d = j + q * TPR;
    const bool ok = d < A;
    pdA[q] = ok ? a.dA[0][(size_t)r * 32 + d] + a.dA[1][(size_t)r * 32 + d] : 0.f;
    pmu[q] = ok ? a.logits_pi[(size_t)r * 2 * A + d] : 0.f;
    praw[q] = ok ? a.logits_pi[(size_t)r * 2 * A + A + d] : 0.f;
    peps[q] = ok ? a.eps_new[(size_t)r * A + d] : 0.f;
    psc[q] = ok ? a.act_scale[d] : 1.f;
  }
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
  const float alpha = a.auto_alpha ? expf(a.log_alpha[0]) : a.alpha_fixed;
  // zero the operand rows (padding included), then fill (dmu | draw)
  for (int e = tid; e < R * 16 * c_out; e += NTHR) xdo[(e / (16 * c_out)) * S.ldx + e % (16 * c_out)] = 0.0f;
  lds_barrier();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int d = j + q * TPR;
    if (d >= A) break;
    const float dA = pdA[q];
    float dmu, draw;
    tanh_gauss_bwd(pmu[q], praw[q], peps[q], psc[q], a.lo_ls, a.hi_ls, dA, alpha * a.inv_B, dmu, draw);
    a.dout_pi[(size_t)r * 2 * A + d] = dmu;
    a.dout_pi[(size_t)r * 2 * A + A + d] = draw;
    a.dout_piT[pk_index(d, r, a.Cb)] = dmu;
    a.dout_piT[pk_index(A + d, r, a.Cb)] = draw;
    a.d_new_act[(size_t)r * A + d] = dA;
    xdo[m * S.ldx + d] = dmu;
    xdo[m * S.ldx + A + d] = draw;
  }
  lds_barrier();
  // ---- policy output layer backward: (dmu | draw) . Wout, then * gelu'(z_last)
  f32x4 acc[RT][4];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  fat_gemm_lds<RT>(acc, a.woutT + (size_t)(4 * wave) * c_out * 256, c_out, 0, c_out, lds, S.off_x + i * S.ldx + 4 * g, S.ldx, lane4);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const f32x4 gq = gload4(a.G[L - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb));
      const f32x4 dz = acc[rt][nt] * gq;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) lds[S.off_h + (16 * rt + 4 * g + rr) * S.ldh + nf[nt]] = dz[rr];
      nt_store4(a.dZ[L - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb), dz);
    }
  lds_barrier();
  const int xs_h = S.off_h + i * S.ldh + 4 * g;
  for (int l = L - 1; l >= 1; --l) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    fat_gemm_lds<RT>(acc, a.wb[l] + (size_t)(4 * wave) * CW * 256, CW, 0, CW, lds, xs_h, S.ldh, lane4);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f32x4 gq = gload4(a.G[l - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb));
        acc[rt][nt] = acc[rt][nt] * gq;
        nt_store4(a.dZ[l - 1] + pk_index(nf[nt], row0 + 16 * rt + 4 * g, a.Cb), acc[rt][nt]);
      }
    if (l > 1) {
      lds_barrier();
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) lds[S.off_h + (16 * rt + 4 * g + rr) * S.ldh + nf[nt]] = acc[rt][nt][rr];
      lds_barrier();
    }
  }
}

template <int NW, int RT>
__global__ void __launch_bounds__(256, 2) k_fat_bwd_pi(BwdPiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_chain_blocks) {
    const int idx = (int)blockIdx.x - a.n_chain_blocks;
    const int per_range = xcd_chunk_grid(a.n_extra);   // n_chain_blocks is a multiple of 8: riders start on XCD 0
    int t;
    if (!xcd_chunk(idx % per_range, a.n_extra, t)) return;
    dw2_tile(a.dw, (idx / per_range) * a.dw.n_base + a.tile0 + t, lds);
    return;
  }
  fat_bwd_pi_body<NW, RT>(a, (int)blockIdx.x, lds);
}

}  // namespace dsact
