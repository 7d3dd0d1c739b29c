// dsact_conv.h -- gfx950 kernels of the CNN encoders (networks/cnn.py:30-53: Conv2d(k, stride, no
// padding) + ReLU stacks; BASELINE.json configs[3]). HIP only, wave64, fp32 matrix cores.
//
// Layout: every activation is pixel-major ("NHWC"): act[m][c], m = (b, y, x). With the patch index
// ordered (ky, kx, ci) one patch ROW (KW*Cin floats) is contiguous in memory, so the implicit-GEMM
// operand  P(m, k) = in[rowoff[m] + k + (k / (KW*Cin)) * (W*Cin - KW*Cin)]  is fetched with the same
// dwordx4 loads as a dense matrix (KW*Cin is a multiple of 4 for every layer of both conv types, the
// first one included: 4*3 and 8*4). Conv weights live in the parameter arena as [Cout][KH][KW][Cin]
// (the Python side exposes them to state_dict as permuted views), i.e. as a dense [Cout x K] matrix.
//
//   forward   Y[m][co]  = relu(sum_k P(m,k) W[co][k] + b[co])           k_conv_fwd  (32x32 tiles, MFMA)
//   weights   dW[co][k] = sum_m dY[m][co] P(m,k),  db[co] = sum_m dY    k_conv_dw   (split over m, MFMA)
//                                                                        k_conv_dw_reduce (+ Adam/Polyak)
//   data      dCol[m][k] = sum_co dY[m][co] W[co][k]                     k_stage<KC,MC,STORE> (dsact_kernels.h)
//             dX[b,y,x,ci] = relu'(x) * sum_{ky,kx} dCol[m(y-ky,x-kx)][ky,kx,ci]   k_col2im (deterministic gather)
#pragma once
#include "dsact_kernels.h"

namespace dsact {

constexpr int kMaxConv = 6;
constexpr int kMaxConvProb = 6;

struct ConvGeom {
  int H, W, Cin, OH, OW, Cout, KS, stride;
  int K;        // KS*KS*Cin
  int KWC;      // KS*Cin: floats of one contiguous patch row
  int rowskip;  // W*Cin - KWC
  float inv_kwc;
};

__device__ __forceinline__ int conv_kmap(const ConvGeom& g, int k) {
  const int ky = (int)(((float)k + 0.5f) * g.inv_kwc);  // exact: k < 2^16, KWC >= 12
  return k + ky * g.rowskip;
}

// exact m / d for 0 <= m < 2^24 (float reciprocal + one correction either way)
__device__ __forceinline__ int fast_div(int m, int d, float inv_d) {
  int q = (int)((float)m * inv_d);
  const int r = m - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}
struct ConvIndex { int OHW; float inv_ohw, inv_ow; };
// float offset of output pixel m's patch origin inside the pixel-major input
__device__ __forceinline__ int conv_rowoff(const ConvGeom& g, const ConvIndex& ix, int m) {
  const int b = fast_div(m, ix.OHW, ix.inv_ohw);
  const int p = m - b * ix.OHW;
  const int oy = fast_div(p, g.OW, ix.inv_ow);
  const int ox = p - oy * g.OW;
  return ((b * g.H + oy * g.stride) * g.W + ox * g.stride) * g.Cin;
}

// A forward "group": up to 3 conv layers of different nets applied to the SAME input (layer 0 of the
// three nets that see `obs`, resp. `obs2`): their weights are concatenated along the channel dimension
// so that one staged patch tile feeds all of them (N = n_sub*Cout). Deeper layers: n_sub = 1.
struct ConvGroup {
  const float* in;       // [B*H*W][Cin]
  const float* w[3];     // [Cout][K] each
  const float* bias[3];
  float* out[3];         // [M][Cout] each
  float* feat[2];        // k_conv_fwd64, last layer of a stack that ends in ONE pixel (round 6): the rows are the feature rows of the
  int ldf;               // MLP trunks too -- also stored at feat[i] + m * ldf (i = 1: the second chain on the same features), or nullptr
  int n_sub;
  int M;
  int tiles_n;           // ceil(n_sub*Cout / 32)
  int item_end;          // exclusive end of this group's work-item range
};
struct ConvStageArgs {
  ConvGeom g;
  ConvIndex ix;
  ConvGroup p[kMaxConvProb];
  int n_prob;
  int n_items;           // total (group, m-tile, n-tile) work items
};

// ---------------------------------------------------------------------------------------------
// (round 3: a flattened version of this loop with three k-tiles of unconditional loads in flight and LDS-only barriers
//  measured SLOWER on layers 2-4 -- 47.8 / 32.0 / 33.4 us vs 39.5 / 26.0 / 31.0 -- at 156 VGPRs = 3 workgroups per CU
//  instead of 4: these tiles are bound by their LDS staging / address arithmetic per 16 MFMAs, not by load latency)
// forward: persistent workgroups; each walks (group, m-tile, n-tile) work items, one 32 (pixels) x 32
// (channels) tile per item. The (item, k-tile) sequence is software pipelined: the global loads of the
// NEXT k-tile (possibly of the next item) are in flight while the current one is multiplied, so the
// L2 round trip is paid once per workgroup, not once per tile.
// ---------------------------------------------------------------------------------------------
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_conv_fwd(ConvStageArgs s) {
  __shared__ __attribute__((aligned(16))) float lds[4 * TILE_LDS];
  const ConvGeom& g = s.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, gq = lane >> 4;
  const int kq = (tid & 15) * 4;
  const int T = (g.K + BK - 1) / BK;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  // per-item state of the loader (rows / weight rows this thread fetches)
  struct Cur { const float* in; const float* q0p; const float* q1p; int ro0, ro1; bool pv0, pv1, qv0, qv1; };
  auto decode = [&](int item, int& pi, int& m0, int& n0) {
    pi = 0;
#pragma unroll
    for (int q = 0; q + 1 < kMaxConvProb; ++q)
      if (q + 1 < s.n_prob && item >= s.p[q].item_end) pi = q + 1;
    const int local = item - (pi ? s.p[pi - 1].item_end : 0);
    const int tn = s.p[pi].tiles_n;
    const int mt = local / tn, nt = local - mt * tn;
    m0 = mt * TM; n0 = nt * TN;
  };
  auto cursor = [&](int item) {
    Cur c;
    int pi, m0, n0;
    decode(item, pi, m0, n0);
    const ConvGroup& t = s.p[pi];
    const int r0 = m0 + (tid >> 4), r1 = r0 + 16;
    c.pv0 = r0 < t.M; c.pv1 = r1 < t.M;
    c.ro0 = conv_rowoff(g, s.ix, c.pv0 ? r0 : t.M - 1);
    c.ro1 = conv_rowoff(g, s.ix, c.pv1 ? r1 : t.M - 1);
    c.in = t.in;
    const int ntot = t.n_sub * g.Cout;
    const int c0 = n0 + (tid >> 4), c1 = c0 + 16;
    c.qv0 = c0 < ntot; c.qv1 = c1 < ntot;
    const int cc0 = c.qv0 ? c0 : 0, cc1 = c.qv1 ? c1 : 0;
    const int s0 = cc0 / g.Cout, s1 = cc1 / g.Cout;
    c.q0p = t.w[s0] + (size_t)(cc0 - s0 * g.Cout) * g.K;
    c.q1p = t.w[s1] + (size_t)(cc1 - s1 * g.Cout) * g.K;
    return c;
  };
  // raw loads only: a select on a just-loaded register would make the wave wait for that load right here and
  // the "prefetch" would not overlap anything -- the masks travel as flags and are applied at LDS-store time
  struct Msk { bool p0, p1, q0, q1; };
  auto load = [&](const Cur& c, int kt, f32x4& P0, f32x4& P1, f32x4& Q0, f32x4& Q1) {
    const int k = kt * BK + kq;
    const bool kv = k < g.K;           // K % 4 == 0: a quad is entirely inside or outside
    const int kc = kv ? k : 0;
    const int km = conv_kmap(g, kc);
    // (quads past K -- 3/4 of the last k-tile at K = 144 -- read ONE fixed address: every distinct line is an L1 request)
    P0 = *(const f32x4u*)(c.in + (kv ? c.ro0 + km : 0));
    P1 = *(const f32x4u*)(c.in + (kv ? c.ro1 + km : 0));
    Q0 = *(const f32x4u*)(kv ? c.q0p + kc : c.in);
    Q1 = *(const f32x4u*)(kv ? c.q1p + kc : c.in);
    Msk m;
    m.p0 = kv && c.pv0; m.p1 = kv && c.pv1; m.q0 = kv && c.qv0; m.q1 = kv && c.qv1;
    return m;
  };
  int item = blockIdx.x;
  if (item >= s.n_items) return;
  Cur cur = cursor(item);
  f32x4 p0, p1, q0, q1;
  Msk mk = load(cur, 0, p0, p1, q0, q1);
  int step = 0;
  while (item < s.n_items) {
    int pi, m0, n0;
    decode(item, pi, m0, n0);
    const ConvGroup& t = s.p[pi];
    // epilogue operands of this item, fetched before the MFMA run
    const int m = m0 + wr * 16 + i;
    const int n = n0 + wc * 16 + 4 * gq;
    const bool in_range = m < t.M && n < t.n_sub * g.Cout;   // Cout % 4 == 0
    const int sub = in_range ? n / g.Cout : 0;
    const int co = n - sub * g.Cout;
    f32x4 bv = zero;
    if (in_range) bv = *(const f32x4u*)(t.bias[sub] + co);
    const int next_item = item + gridDim.x;
    Cur nxt = cur;
    f32x4 acc0 = zero, acc1 = zero;
    for (int kt = 0; kt < T; ++kt, ++step) {
      float* Ps = lds + (step & 1) * 2 * TILE_LDS;
      float* Qs = Ps + TILE_LDS;
      if (!mk.p0) p0 = zero;
      if (!mk.p1) p1 = zero;
      if (!mk.q0) q0 = zero;
      if (!mk.q1) q1 = zero;
      tile_store_lds<false>(Ps, tid, p0, p1);
      tile_store_lds<false>(Qs, tid, q0, q1);
      __syncthreads();
      if (kt + 1 < T) mk = load(cur, kt + 1, p0, p1, q0, q1);
      else if (next_item < s.n_items) { nxt = cursor(next_item); mk = load(nxt, 0, p0, p1, q0, q1); }
      tile_mma<false, false>(Ps, Qs, wr * 16 + i, wc * 16 + i, gq, acc0, acc1);
    }
    if (in_range) {
      f32x4 o = acc0 + acc1 + bv;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
      *(f32x4u*)(t.out[sub] + (size_t)m * g.Cout + co) = o;
    }
    cur = nxt;
    item = next_item;
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// forward, wide layers (round 4): 64 (pixels) x 64 (channels) tiles, 512 threads -- run_tile64's structure (dsact_kernels.h:
// 8 waves, wave (wr, wc) owns rows wr*32..+31 x columns wc*16..+15, k-tiles of 64 double-buffered through LDS with the next
// tile's loads in flight) with the patch operand gathered through conv_rowoff / conv_kmap. Half the operand bytes per FLOP
// of the 32 x 32 tiles above; layers with M % 64 == 0, Cout % 64 == 0 and one net per group (type_2 layers 3, 4).
// ---------------------------------------------------------------------------------------------
// MB = 16-row blocks per wave: 2 = 64-pixel tiles; 1 = 32 (pixels) x 64 (channels) tiles, k-tiles of 64 -- half the dependent
// k-tile steps of the 32 x 32 kernel for a layer whose 64 x 64 tiles would not fill the chip (type_2 layer 5: M = batch)
template <int MB>
__global__ void __launch_bounds__(kThreads64) k_conv_fwd64(ConvStageArgs s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const ConvGeom& g = s.g;
  const int b = (int)blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int q = 0; q + 1 < kMaxConvProb; ++q)
    if (q + 1 < s.n_prob && b >= s.p[q].item_end) pi = q + 1;
  const ConvGroup& t = s.p[pi];
  const int local = b - (pi ? s.p[pi - 1].item_end : 0);
  const int tn = g.Cout >> 6;
  const int mt = local / tn, nt = local - mt * tn;
  const int m0 = mt * 32 * MB, n0 = nt * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int i = lane & 15, gq = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const int T = (g.K + BK - 1) / BK;
  // this thread's two patch rows (pixels) and two weight rows (channels) of every k-tile; k quad (tid & 15)
  const float* pa = t.in + conv_rowoff(g, s.ix, m0 + (tid >> 4));
  const float* pb = MB == 2 ? t.in + conv_rowoff(g, s.ix, m0 + (tid >> 4) + 32) : pa;   // (MB == 1: the second row slot is unused)
  const float* qa = t.w[0] + (size_t)(n0 + (tid >> 4)) * g.K;
  const float* qb = qa + (size_t)32 * g.K;
  auto load = [&](int kt, f32x4& p0, f32x4& p1, f32x4& q0, f32x4& q1) {
    const int k = kt * BK + (tid & 15) * 4;
    const int kc = k < g.K ? k : 0;             // (K % 4 == 0: a quad is inside or outside; outside: masked at store time)
    const int km = conv_kmap(g, kc);
    p0 = *(const f32x4u*)(pa + km); p1 = *(const f32x4u*)(pb + km);
    q0 = *(const f32x4u*)(qa + kc); q1 = *(const f32x4u*)(qb + kc);
  };
  const int n = n0 + wc * 16 + 4 * gq;
  const f32x4 bv = *(const f32x4u*)(t.bias[0] + n);
  f32x4 acc[2] = {zero, zero};
  f32x4 p0, p1, q0, q1;
  load(0, p0, p1, q0, q1);
  for (int kt = 0; kt < T; ++kt) {
    float* Ps = lds + (kt & 1) * 2 * TILE64_LDS;
    float* Qs = Ps + TILE64_LDS;
    if (!(kt * BK + (tid & 15) * 4 < g.K)) { p0 = zero; p1 = zero; q0 = zero; q1 = zero; }
    tile64_store_lds<false>(Ps, tid, p0, p1);
    tile64_store_lds<false>(Qs, tid, q0, q1);
    __syncthreads();
    if (kt + 1 < T) load(kt + 1, p0, p1, q0, q1);
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const f32x4 q = frag_read(Qs, wc * 16 + i, kk, gq);
      const f32x4 xa = frag_read(Ps, wr * 16 * MB + i, kk, gq);
      const f32x4 xb = MB == 2 ? frag_read(Ps, wr * 32 + 16 + i, kk, gq) : xa;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], xa[e], acc[0], 0, 0, 0);
        if (MB == 2) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(q[e], xb[e], acc[1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m0 + wr * 16 * MB + mb * 16 + i;
    f32x4 o = acc[mb] + bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
    *(f32x4u*)(t.out[0] + (size_t)m * g.Cout + n) = o;
    if (t.feat[0]) *(f32x4u*)(t.feat[0] + (size_t)m * t.ldf + n) = o;   // (uniform per group: what k_feat_scatter did for P == 1)
    if (t.feat[1]) *(f32x4u*)(t.feat[1] + (size_t)m * t.ldf + n) = o;
  }
}

// ---------------------------------------------------------------------------------------------
// forward, narrow layers (K <= 16*NKK <= 80, n_sub*Cout <= 32: the first two layers of conv type_2, which
// hold 2/3 of the stack's pixels): WAVE-autonomous tiles, no LDS, no barriers. A lane (i = lane&15,
// g = lane>>4) feeds the matrix core with row i of both operands and the four k = 16*kk + 4*g + e of MFMA
// e -- exactly one dwordx4 of a patch row / weight row, so fragments are loaded straight from L2 into
// registers. The weight fragments of a group stay in registers across all of the wave's tiles; the patch
// fragments of the next tile are in flight while the current 32 x 32 tile is multiplied.
// ---------------------------------------------------------------------------------------------
template <int NKK, int NB>   // k groups of 16, channel blocks of 16
__global__ void __launch_bounds__(kThreads) k_conv_fwd_narrow(ConvStageArgs s) {
  const ConvGeom& g = s.g;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, gq = lane >> 4;
  const int wave_global = __builtin_amdgcn_readfirstlane(blockIdx.x * (kThreads / 64) + (threadIdx.x >> 6));
  // a wave works on ONE group (s.n_items = waves per group here): its weight fragments are loaded once, before
  // the tile loop, so that nothing but patch loads, MFMAs and stores remains inside it
  const int wpg = s.n_items;
  const int pi = wave_global / wpg;
  if (pi >= s.n_prob) return;
  const int wv = wave_global - pi * wpg;
  const ConvGroup& t = s.p[pi];
  const int ntot = t.n_sub * g.Cout;
  const int n_tiles = (t.M + 31) >> 5;
  if (wv >= n_tiles) return;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  int km[NKK];
#pragma unroll
  for (int kk = 0; kk < NKK; ++kk) {
    const int k = 16 * kk + 4 * gq;
    km[kk] = conv_kmap(g, k < g.K ? k : 0);
  }
  // weights: zero beyond K (the patch operand is NOT masked: whatever it holds there multiplies 0) and beyond the
  // group's channels
  f32x4 Q[NB][NKK], bv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 16 + i;
    const bool nv = n < ntot;
    const int nc = nv ? n : 0;
    const int sub = nc / g.Cout;
    const float* wrow = t.w[sub] + (size_t)(nc - sub * g.Cout) * g.K;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
      const bool kvv = 16 * kk + 4 * gq < g.K;
      Q[nb][kk] = *(const f32x4u*)(wrow + (kvv ? 16 * kk + 4 * gq : 0));
      if (!(nv && kvv)) Q[nb][kk] = zero;
    }
    const int nq = nb * 16 + 4 * gq;   // this lane's output channels of block nb
    const bool qv = nq < ntot;
    const int sq = qv ? nq / g.Cout : 0;
    bv[nb] = qv ? *(const f32x4u*)(t.bias[sq] + (nq - sq * g.Cout)) : zero;
  }
  // output addressing of this lane (constant across tiles)
  // (the output pointer is selected HERE from three scalar loads: `t.out[o_sub]` with a per-lane index inside the tile
  //  loop was a vector load of the pointer + s_waitcnt vmcnt(0) in front of every store -- the next tile's patch loads,
  //  which are supposed to be in flight during the MFMAs, were drained there; round 3, disassembly)
  typedef __attribute__((address_space(4))) const ConvStageArgs KArgs;
  KArgs* ka = (KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  float* const out0 = ka->p[pi].out[0]; float* const out1 = ka->p[pi].out[1]; float* const out2 = ka->p[pi].out[2];
  int o_co[NB]; bool o_ok[NB]; float* o_ptr[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = nb * 16 + 4 * gq;
    o_ok[nb] = n < ntot;
    const int o_sub = o_ok[nb] ? n / g.Cout : 0;
    o_co[nb] = n - o_sub * g.Cout;
    o_ptr[nb] = o_sub == 0 ? out0 : (o_sub == 1 ? out1 : out2);
  }
  // RAW patch loads (rows clamped into the matrix, nothing selected on the loaded registers: a select would make
  // the wave wait for the load at issue time and the prefetch would overlap nothing). The loader visits the wave's tiles
  // in order (tile wv, wv + wpg, ...: 32 * wpg pixels apart) and keeps (ox, oy, offset) of its two rows incrementally --
  // see k_conv_dw: two reciprocal divisions + five quarter-rate multiplies per row and tile were a quarter of a tile's
  // 24-48 MFMAs.
  const int SX = g.stride * g.Cin, SY = g.stride * g.W * g.Cin, SB = g.H * g.W * g.Cin;
  const int adv = 32 * wpg;                                       // pixels between consecutive tiles of this wave (uniform)
  const int adb = adv / s.ix.OHW, adr = adv - adb * s.ix.OHW, ady = adr / g.OW, adx = adr - ady * g.OW;
  const int adoff = adb * SB + ady * SY + adx * SX, wrapx = SY - g.OW * SX, wrapy = SB - g.OH * SY;
  const int off_last = conv_rowoff(g, s.ix, t.M - 1);
  int pm[2], pox[2], poy[2], poff[2];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    const int m = wv * 32 + mb * 16 + i;
    const int mc_ = m < t.M ? m : t.M - 1;
    const int b_ = fast_div(mc_, s.ix.OHW, s.ix.inv_ohw);
    const int p_ = mc_ - b_ * s.ix.OHW;
    poy[mb] = fast_div(p_, g.OW, s.ix.inv_ow);
    pox[mb] = p_ - poy[mb] * g.OW;
    poff[mb] = b_ * SB + poy[mb] * SY + pox[mb] * SX;
    pm[mb] = m;
  }
  // loads the loader's current tile, then moves it one tile on (rows past the end read the last pixel's patch)
  auto load_next = [&](f32x4 (&P)[2][NKK]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float* base = t.in + (pm[mb] < t.M ? poff[mb] : off_last);
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) P[mb][kk] = *(const f32x4u*)(base + km[kk]);
      pm[mb] += adv;
      pox[mb] += adx; poff[mb] += adoff;
      if (pox[mb] >= g.OW) { pox[mb] -= g.OW; poy[mb] += 1; poff[mb] += wrapx; }
      poy[mb] += ady;
      if (poy[mb] >= g.OH) { poy[mb] -= g.OH; poff[mb] += wrapy; }
    }
  };
  auto compute = [&](int tile, const f32x4 (&P)[2][NKK]) {
    f32x4 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = zero;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Q[nb][kk][e], P[mb][kk][e], acc[mb][nb], 0, 0, 0);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int m = tile * 32 + mb * 16 + i;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        if (m < t.M && o_ok[nb]) {
          f32x4 o = acc[mb][nb] + bv[nb];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
          *(f32x4u*)(o_ptr[nb] + (size_t)m * g.Cout + o_co[nb]) = o;
        }
      }
    }
  };
  f32x4 Pa[2][NKK], Pb[2][NKK];
  load_next(Pa);
  for (int tile = wv; tile < n_tiles; tile += 2 * wpg) {
    const int nx = tile + wpg;
    load_next(Pb);                            // unconditional (clamped inside): no branch between the loads and the MFMAs
    // (round 6: hipcc sinks these requests behind all but the tile's last 4-14 MFMAs -- the next tile's patches are in flight
    //  under the stores, not under the MFMAs. The one-wave-per-SIMD form pins them where they are written: layer 2 27.7 ->
    //  25.9 us. The narrower forms, 3-4 waves per SIMD on HBM-bound layers, LOSE with the pin -- layer 0 49.3 -> 59.8 us, layer
    //  1 33.5 -> 41.0: the other waves cover the latency, and 4 waves' requests at once queue behind each other;
    //  profiles/r06_isa_schedule_fixes.txt)
    if (NKK >= 9) __builtin_amdgcn_sched_barrier(0);
    compute(tile, Pa);
    if (nx >= n_tiles) break;
    load_next(Pa);
    if (NKK >= 9) __builtin_amdgcn_sched_barrier(0);
    compute(nx, Pb);
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient, split over the pixel dimension: workgroup (problem, chunk, co-tile, k-tile) forms
//   part[chunk][co][kidx] = sum_{m in chunk} dY[m][co] * P'(m, kidx),  kidx in [0, K]  (P'(m, K) = 1 -> bias)
// Both operands are "row contiguous" in memory (dY: co contiguous, patch: k contiguous) and are
// transposed into the k-contiguous LDS image on the way in (tile_store_lds<true>).
// Like the forward groups, a problem may carry up to 3 dY's that share the input (layer 0 of the nets
// that see `obs`): their channel rows are concatenated so that the gathered patch tile is staged once.
// ---------------------------------------------------------------------------------------------
struct ConvDwProb {
  const float* in;       // layer input  [B*H*W][Cin]
  const float* dy[3];    // [M][Cout] each
  float* part[3];        // [n_chunks][Cout][K1p] each
  int n_sub;
  int M;
  int tiles_co;          // ceil(n_sub*Cout / 32)
  int block_end;         // exclusive end of this problem's block range
};
struct ConvDwArgs {
  ConvGeom g;
  ConvIndex ix;          // patch origins are computed (two exact reciprocal divisions), not looked up: a table
                         // lookup is a dependent load in front of every gathered load
  ConvDwProb p[3];
  int n_prob;
  int chunk;          // pixels per chunk (multiple of 64)
  int n_chunks, tiles_k;  // per problem: n_chunks * tiles_co * tiles_k blocks
  int K1p;            // padded partial row length: K + 4
};

// NKT = k-tiles (32 patch columns each) per workgroup: the dY tile of a step is staged once and multiplied
// with NKT patch tiles, so dY is not re-read per k-tile (PMC: the weight gradient was the largest consumer
// of memory-side traffic, 2-3x its algorithmic bytes, when every k-tile had its own workgroup).
// NS = steps of global loads in flight (register sets).
// DB = LDS double buffering (one barrier per step, 2 x (1 + NKT) tiles); false: ONE set of 1 + NKT tiles and a second barrier per
// step -- NKT = 3 then costs the LDS of the double-buffered NKT = 1 form (four workgroups per CU) while dY is read once.
template <int NKT, int NS = 2, bool DB = true>
__global__ void __launch_bounds__(kThreads) k_conv_dw(ConvDwArgs s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // (DB ? 2 : 1) x (1 + NKT) tiles
  const int b = blockIdx.x;
  int pi = 0;
  if (s.n_prob > 1 && b >= s.p[0].block_end) pi = 1;
  if (s.n_prob > 2 && b >= s.p[1].block_end) pi = 2;
  const ConvDwProb& t = s.p[pi];
  const ConvGeom& g = s.g;
  int local = b - (pi ? s.p[pi - 1].block_end : 0);
  const int per_chunk = t.tiles_co * s.tiles_k;   // tiles_k counts k-GROUPS of NKT tiles here
  const int ch = local / per_chunk;
  local -= ch * per_chunk;
  const int ct = local / s.tiles_k, kg = local - ct * s.tiles_k;
  const int co0 = ct * TM, k0 = kg * NKT * TN;
  const int mb = ch * s.chunk;
  const int me = mb + s.chunk < t.M ? mb + s.chunk : t.M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, gq = lane >> 4;
  const int ntot = t.n_sub * g.Cout;
  // MC cursors: this thread's 4 consecutive rows (co / kidx) and its two m slots per 64-pixel tile
  const int pc = co0 + (tid & 7) * 4;            // (concatenated) co quad; Cout % 4 == 0: one sub per quad
  const bool pcv = pc < ntot;
  const int psub = pcv ? pc / g.Cout : 0;
  const float* dyp = t.dy[psub] + (pcv ? pc - psub * g.Cout : 0);
  int qmode[NKT], qmap[NKT];
#pragma unroll
  for (int u = 0; u < NKT; ++u) {
    const int qk = k0 + u * TN + (tid & 7) * 4;  // kidx quad of k-tile u
    qmode[u] = qk < g.K ? 0 : (qk == g.K ? 1 : 2);  // gathered / bias column / padding
    qmap[u] = conv_kmap(g, qmode[u] == 0 ? qk : 0);
  }
  const int ms = tid >> 3;                       // m slot 0 (slot 1 = +32)
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const f32x4 one0 = {1.f, 0.f, 0.f, 0.f};
  // raw loads; the masks (pixel validity of the two slots) are applied at LDS-store time -- see k_conv_fwd.
  // The loader walks its two pixel slots 64 pixels per step INCREMENTALLY: (ox, oy, float offset of the patch origin) are
  // advanced by the decomposition of 64 = db*OH*OW + doy*OW + dox with two compare / subtract carries, instead of two
  // reciprocal divisions + five integer multiplies per slot and step -- v_mul_lo_u32 is quarter rate and f32 VALU work
  // shares the lanes with the MFMAs: the address arithmetic was as long as the step's 16 MFMAs (disassembly, round 3).
  const int SX = g.stride * g.Cin, SY = g.stride * g.W * g.Cin, SB = g.H * g.W * g.Cin;
  const int db64 = BK / s.ix.OHW, r64 = BK - db64 * s.ix.OHW, doy64 = r64 / g.OW, dox64 = r64 - doy64 * g.OW;   // uniform
  const int doff64 = db64 * SB + doy64 * SY + dox64 * SX, wrapx = SY - g.OW * SX, wrapy = SB - g.OH * SY;
  const int off_last = conv_rowoff(g, s.ix, t.M - 1);
  const size_t dy_last = (size_t)(t.M - 1) * g.Cout;
  struct Pix { int m, ox, oy, off; };
  auto pix_at = [&](int m) {
    Pix px;
    const int mc_ = m < t.M ? m : t.M - 1;     // (decomposed once per block; steps past the end are masked by m >= me)
    const int b_ = fast_div(mc_, s.ix.OHW, s.ix.inv_ohw);
    const int p_ = mc_ - b_ * s.ix.OHW;
    px.oy = fast_div(p_, g.OW, s.ix.inv_ow);
    px.ox = p_ - px.oy * g.OW;
    px.off = b_ * SB + px.oy * SY + px.ox * SX;
    px.m = m;
    return px;
  };
  auto pix_next = [&](Pix& px) {
    px.m += BK;
    px.ox += dox64; px.off += doff64;
    if (px.ox >= g.OW) { px.ox -= g.OW; px.oy += 1; px.off += wrapx; }
    px.oy += doy64;
    if (px.oy >= g.OH) { px.oy -= g.OH; px.off += wrapy; }
  };
  Pix pxa = pix_at(mb + ms), pxc = pix_at(mb + ms + 32);
  auto load_next = [&](f32x4& P0, f32x4& P1, f32x4 (&Q0)[NKT], f32x4 (&Q1)[NKT]) {
    const bool oka = pxa.m < me, okc = pxc.m < me;
    const int roa = oka ? pxa.off : off_last, roc = okc ? pxc.off : off_last;
    // Lanes whose quad lies outside the operand (channel quads past n_sub*Cout, the bias / padding columns of the last
    // k-tile -- HALF the lanes on the 8- and 16-channel layers) read ONE fixed address instead of a row-dependent one:
    // what they load is replaced at LDS-store time, but every distinct cache line a wave-load touches is a request to
    // the CU's L1, and these 16-bytes-per-lane gathers are bound by exactly that request rate.
    P0 = *(const f32x4u*)(dyp + (pcv ? (oka ? (size_t)pxa.m * g.Cout : dy_last) : 0));
    P1 = *(const f32x4u*)(dyp + (pcv ? (okc ? (size_t)pxc.m * g.Cout : dy_last) : 0));
#pragma unroll
    for (int u = 0; u < NKT; ++u) {
      Q0[u] = *(const f32x4u*)(t.in + (qmode[u] == 0 ? roa + qmap[u] : 0));
      Q1[u] = *(const f32x4u*)(t.in + (qmode[u] == 0 ? roc + qmap[u] : 0));
    }
    pix_next(pxa); pix_next(pxc);
  };
  auto mask = [&](int it, f32x4& P0, f32x4& P1, f32x4 (&Q0)[NKT], f32x4 (&Q1)[NKT]) {
    const int ma = mb + it * BK + ms, mc = ma + 32;
    if (!(ma < me && pcv)) P0 = zero;            // masking dY is enough: the other operand is finite
    if (!(mc < me && pcv)) P1 = zero;
#pragma unroll
    for (int u = 0; u < NKT; ++u) {
      if (qmode[u] == 1) { Q0[u] = one0; Q1[u] = one0; }
      else if (qmode[u] == 2) { Q0[u] = zero; Q1[u] = zero; }
    }
  };
  const int T = (me - mb + BK - 1) / BK;
  f32x4 acc0[NKT], acc1[NKT];
#pragma unroll
  for (int u = 0; u < NKT; ++u) { acc0[u] = zero; acc1[u] = zero; }
  // NS steps of global loads in flight (register sets rotate): one step's L2/MALL round trip is longer than its
  // 16 MFMAs + LDS staging, and a chunk is a dependent chain of 4-16 steps. The refill of a set is UNCONDITIONAL (step index
  // clamped to the last one): with a branch around it hipcc's wait-count pass falls back to s_waitcnt vmcnt(0) at the
  // next use -- every step then waited for the loads issued one step earlier (found in the disassembly, round 3).
  f32x4 ps0[NS], ps1[NS], qs0[NS][NKT], qs1[NS][NKT];
#pragma unroll
  for (int st = 0; st < NS; ++st) load_next(ps0[st], ps1[st], qs0[st], qs1[st]);   // steps 0 .. NS-1 (past T: masked, clamped)
  auto step = [&](int it, f32x4& P0, f32x4& P1, f32x4 (&Q0)[NKT], f32x4 (&Q1)[NKT]) {
    float* Ps = lds + (DB ? (it & 1) : 0) * (1 + NKT) * TILE_LDS;
    mask(it, P0, P1, Q0, Q1);
    tile_store_lds<true>(Ps, tid, P0, P1);
#pragma unroll
    for (int u = 0; u < NKT; ++u) tile_store_lds<true>(Ps + (1 + u) * TILE_LDS, tid, Q0[u], Q1[u]);
    lds_barrier();
    load_next(P0, P1, Q0, Q1);                 // step it + NS of the loader
#pragma unroll
    for (int u = 0; u < NKT; ++u)
      tile_mma<true, true>(Ps, Ps + (1 + u) * TILE_LDS, wr * 16 + i, wc * 16 + i, gq, acc0[u], acc1[u]);
    if (!DB) lds_barrier();                    // the next step's stores overwrite these tiles
  };
  int it = 0;
  for (; it + NS <= T; it += NS) {
#pragma unroll
    for (int st = 0; st < NS; ++st) step(it + st, ps0[st], ps1[st], qs0[st], qs1[st]);
  }
  // tail (T % NS steps): the sets hold steps it, it + 1, ... in order
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (it + st < T) step(it + st, ps0[st], ps1[st], qs0[st], qs1[st]);
  const int cot = co0 + wr * 16 + i;
  if (cot < ntot) {
    const int sub = cot / g.Cout, co = cot - sub * g.Cout;
#pragma unroll
    for (int u = 0; u < NKT; ++u) {
      const int kk = k0 + u * TN + wc * 16 + 4 * gq;
      if (kk < s.K1p) *(f32x4u*)(t.part[sub] + ((size_t)ch * g.Cout + co) * s.K1p + kk) = acc0[u] + acc1[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight gradient of the narrow layers (n_sub * Cout <= 16 * NCB, K + 4 <= 16 * NKB) on REGISTER tiles: no LDS staging, no
// transposition, no barrier inside the contraction (round 4; the LDS-tile kernel above spends a step on its transposing
// stores and 16-byte gathers, not on bytes or MFMAs -- DESIGN.md 4b). With the contraction over pixels,
// v_mfma_f32_16x16x4_f32 wants from lane (i = lane & 15, g = lane >> 4)  A[i][g] = dY[px0 + g][co0 + i]  and
// B[g][i] = P[px0 + g][k0 + i]: both are coalesced 4-byte loads from the row-major operands as they lie in memory.
// D[co0 + 4g + r][k0 + i] accumulates in NCB x NKB register tiles per wave. A workgroup = one pixel chunk of one problem,
// its four waves split the chunk's pixels; 16 pixels (4 MFMA steps) of loads are in flight while the previous 16 are
// multiplied; the four partial tiles meet in LDS once, at the end (fixed order). Same partial layout as k_conv_dw.
// Measured at type_2, batch 256 (two groups in flight, ~512 workgroups): layer 1 37.0 -> 30.3 us, layer 0 33.4 -> 41.0, layer 2
// (228 VGPRs, one wave per SIMD) 20.2 -> 37.0; more, shorter workgroups are slower (layer 1: 39 us at 1024-2048): chosen per
// layer (DSACT_CONV_DW_REG = bit mask of layers; default: the layers with 16 channels and three k-tiles, i.e. layer 1).
// ---------------------------------------------------------------------------------------------
template <int NCB, int NKB>
__global__ void __launch_bounds__(kThreads) k_conv_dw_reg(ConvDwArgs s) {
  __shared__ f32x4 red[NCB * NKB][64];
  const ConvGeom& g = s.g;
  const int b = blockIdx.x;
  int pi = 0;
  if (s.n_prob > 1 && b >= s.p[0].block_end) pi = 1;
  if (s.n_prob > 2 && b >= s.p[1].block_end) pi = 2;
  const ConvDwProb& t = s.p[pi];
  const int ch = b - (pi ? s.p[pi - 1].block_end : 0);
  const int mb = ch * s.chunk;
  const int me = mb + s.chunk < t.M ? mb + s.chunk : t.M;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 15, gq = lane >> 4;
  const int ntot = t.n_sub * g.Cout;
  // this wave's pixels: a quarter of the chunk, rounded up to whole groups of 16
  const int per = (((me - mb + 3) >> 2) + 15) & ~15;
  const int w0 = mb + wave * per;
  const int w1 = w0 + per < me ? w0 + per : me;
  // operand addressing of this lane
  const float* dyp[NCB]; bool cov[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int co = cb * 16 + i;
    cov[cb] = co < ntot;
    const int sub = cov[cb] ? co / g.Cout : 0;
    dyp[cb] = t.dy[sub] + (cov[cb] ? co - sub * g.Cout : 0);
  }
  int km[NKB], kmode[NKB];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int k = kb * 16 + i;
    kmode[kb] = k < g.K ? 0 : (k == g.K ? 1 : 2);   // gathered / bias column (constant 1) / padding
    km[kb] = conv_kmap(g, kmode[kb] == 0 ? k : 0);
  }
  // pixel walk of this lane: pixel w0 + gq, then 4 pixels per MFMA step (ox, oy, patch origin advanced incrementally)
  const int SX = g.stride * g.Cin, SY = g.stride * g.W * g.Cin, SB = g.H * g.W * g.Cin;
  const int d4y = 4 / g.OW, d4x = 4 - d4y * g.OW;           // OW >= 2: at most ... (host: OW >= 4, so d4y = 0 or 1)
  const int doff4 = d4y * SY + d4x * SX, wrapx = SY - g.OW * SX, wrapy = SB - g.OH * SY;
  const int off_last = conv_rowoff(g, s.ix, t.M - 1);
  int pm = w0 + gq, pox, poy, poff;
  {
    const int mc_ = pm < t.M ? pm : t.M - 1;
    const int b_ = fast_div(mc_, s.ix.OHW, s.ix.inv_ohw);
    const int p_ = mc_ - b_ * s.ix.OHW;
    poy = fast_div(p_, g.OW, s.ix.inv_ow);
    pox = p_ - poy * g.OW;
    poff = b_ * SB + poy * SY + pox * SX;
  }
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  struct Grp { float a[NCB][4]; float bq[NKB][4]; };
  auto load_grp = [&](Grp& G) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = pm < w1;
      const size_t ao = ok ? (size_t)pm * g.Cout : (size_t)(t.M - 1) * g.Cout;
      const int ro = ok ? poff : off_last;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) G.a[cb][e] = dyp[cb][cov[cb] ? ao : 0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) G.bq[kb][e] = t.in[kmode[kb] == 0 ? ro + km[kb] : 0];
      pm += 4;
      pox += d4x; poy += d4y; poff += doff4;
      if (pox >= g.OW) { pox -= g.OW; poy += 1; poff += wrapx; }
      if (poy >= g.OH) { poy -= g.OH; poff += wrapy; }
    }
  };
  f32x4 acc[NCB][NKB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) acc[cb][kb] = zero;
  auto compute = [&](int p0, const Grp& G) {    // p0: first pixel of the group (lane's pixels: p0 + gq + 4e)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool ok = p0 + gq + 4 * e < w1;
      float av[NCB], bv[NKB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) av[cb] = (ok && cov[cb]) ? G.a[cb][e] : 0.0f;    // masking dY is enough: the other operand is finite
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) bv[kb] = kmode[kb] == 0 ? G.bq[kb][e] : (kmode[kb] == 1 ? 1.0f : 0.0f);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) acc[cb][kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[cb], bv[kb], acc[cb][kb], 0, 0, 0);
    }
  };
  // NG groups of 16 pixels in flight: a group's 4 x NCB x NKB MFMAs are 0.3-0.4 us, a memory round trip is several times that.
  // Everything is unconditional (loads clamped, masked groups multiply zeros): no branch between the loads and the MFMAs.
  constexpr int NG = 2;   // (4: layer 1 33.0 instead of 30.3 us -- the registers cost more occupancy than the depth hides)
  Grp G[NG];
#pragma unroll
  for (int u = 0; u < NG; ++u) load_grp(G[u]);
  for (int p0 = w0; p0 < w1; p0 += 16 * NG) {
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      compute(p0 + 16 * u, G[u]);
      load_grp(G[u]);                    // the group NG further on
    }
  }
  // the four waves' partial tiles, added in wave order
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) red[cb * NKB + kb][lane] = acc[cb][kb];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) acc[cb][kb] += red[cb * NKB + kb][lane];
    }
    __syncthreads();
  }
  if (wave != 0) return;
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cot = cb * 16 + 4 * gq + r;
      if (cot >= ntot) continue;
      const int sub = cot / g.Cout, co = cot - sub * g.Cout;
      float* dst = t.part[sub] + ((size_t)ch * g.Cout + co) * s.K1p;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int kk = kb * 16 + i;
        if (kk < s.K1p) dst[kk] = acc[cb][kb][r];
      }
    }
}

// ---------------------------------------------------------------------------------------------
// data gradient of the narrow layers (Cin <= 16) without a column buffer:
//   dX[b,y,x,:] = relu'(x) * sum_{ky,kx valid} sum_co dY[b,(y-ky)/s,(x-kx)/s,co] * W[co][ky][kx][:]
// one thread per input pixel, all Cin channels in registers. A workgroup handles pixels of ONE parity
// class (y mod s, x mod s), so the set of contributing taps -- and with it every weight address -- is
// wave-uniform: weights arrive through the scalar cache and feed v_fmac as SGPR operands; the only vector
// loads are the dY rows. These layers have a huge pixel count and 8-16 channels: matrix-core tiles would be
// mostly padding, and the [M x K] column buffer of the dCol/col2im route is 4.5x the size of dX.
// ---------------------------------------------------------------------------------------------
struct ConvDxArgs {
  ConvGeom g;
  const float* dy[3];     // [M][Cout]
  const float* w[3];      // [Cout][K]
  const float* x[3];      // layer input activations [B*H*W][Cin] (ReLU mask)
  float* dx[3];           // [B*H*W][Cin]
  int n_prob;
  int B;
};
template <int NQ>   // Cin / 4
__global__ void __launch_bounds__(kThreads) k_conv_dx_direct(ConvDxArgs a) {
  const ConvGeom& g = a.g;
  const int pi = blockIdx.y;
  const int st = g.stride;
  const int py = blockIdx.z / st, px = blockIdx.z - py * st;
  const int Yq = (g.H - py + st - 1) / st, Xq = (g.W - px + st - 1) / st;
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= a.B * Yq * Xq) return;
  const int xq = idx % Xq;
  const int t1 = idx / Xq;
  const int yq = t1 % Yq, b = t1 / Yq;
  f32x4 acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* __restrict__ dyb = a.dy[pi];
  const float* __restrict__ wb = a.w[pi];
  for (int ky = py, ay = 0; ky < g.KS; ky += st, ++ay) {
    const int oy = yq - ay;
    const bool vy = oy >= 0 && oy < g.OH;
    for (int kx = px, ax = 0; kx < g.KS; kx += st, ++ax) {
      const int ox = xq - ax;
      const bool v = vy && ox >= 0 && ox < g.OW;
      const float* dyp = dyb + (((size_t)b * g.OH + (v ? oy : 0)) * g.OW + (v ? ox : 0)) * g.Cout;
      const float* wt = wb + (ky * g.KS + kx) * g.Cin;   // uniform across the workgroup
      for (int c4 = 0; c4 < g.Cout; c4 += 4) {
        f32x4 d = *(const f32x4u*)(dyp + c4);
        if (!v) d = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* wr_ = wt + (size_t)(c4 + e) * g.K;
          const f32x4 dv = {d[e], d[e], d[e], d[e]};
#pragma unroll
          for (int q = 0; q < NQ; ++q)   // explicit fma (the build disables contraction): packed v_pk_fma_f32
            acc[q] = __builtin_elementwise_fma(dv, *(const f32x4u*)(wr_ + 4 * q), acc[q]);
        }
      }
    }
  }
  const int y = yq * st + py, x = xq * st + px;
  const size_t o = (((size_t)b * g.H + y) * g.W + x) * (4 * NQ);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const f32x4 xv = *(const f32x4u*)(a.x[pi] + o + 4 * q);
    f32x4 r = acc[q];
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = xv[e] > 0.f ? r[e] : 0.f;
    *(f32x4u*)(a.dx[pi] + o + 4 * q) = r;
  }
}

// Same contraction, one thread per S x S block of input pixels (all stride-parity classes at once): the block's
// pixels share the (KS/S)^2 neighbouring dY rows, which are loaded once, and the thread writes S adjacent pixels
// per image row -- whole cache lines across the wave instead of every other pixel.
template <int NQ, int KS, int S>
__global__ void __launch_bounds__(kThreads) k_conv_dx_block(ConvDxArgs a) {
  constexpr int AY = (KS + S - 1) / S;
  const ConvGeom& g = a.g;
  const int pi = blockIdx.y;
  const int Yb = (g.H + S - 1) / S, Xb = (g.W + S - 1) / S;
  const int idx0 = blockIdx.x * kThreads + threadIdx.x;
  const bool live = idx0 < a.B * Yb * Xb;       // dead threads compute on clamped indices and store nothing
  const int idx = live ? idx0 : 0;
  const int xq = idx % Xb;
  const int t1 = idx / Xb;
  const int yq = t1 % Yb, b = t1 / Yb;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[S * S][NQ];
#pragma unroll
  for (int c = 0; c < S * S; ++c)
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[c][q] = zero;
  const float* __restrict__ wb = a.w[pi];
  const float* dyp[AY][AY];
  bool dv[AY][AY];
#pragma unroll
  for (int ay = 0; ay < AY; ++ay)
#pragma unroll
    for (int ax = 0; ax < AY; ++ax) {
      const int oy = yq - ay, ox = xq - ax;
      dv[ay][ax] = oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
      dyp[ay][ax] = a.dy[pi] + (((size_t)b * g.OH + (dv[ay][ax] ? oy : 0)) * g.OW + (dv[ay][ax] ? ox : 0)) * g.Cout;
    }
  for (int c4 = 0; c4 < g.Cout; c4 += 4) {
    // measured slower: all 16 channels' quads of a trip up front (118-168 VGPRs, spilled SGPRs), and a ping-pong
    // prefetch of the next channel quad (34 -> 42 us on layer 1)
    f32x4 d[AY][AY];
#pragma unroll
    for (int ay = 0; ay < AY; ++ay)
#pragma unroll
      for (int ax = 0; ax < AY; ++ax) d[ay][ax] = *(const f32x4u*)(dyp[ay][ax] + c4);
#pragma unroll
    for (int ay = 0; ay < AY; ++ay)
#pragma unroll
      for (int ax = 0; ax < AY; ++ax) if (!dv[ay][ax]) d[ay][ax] = zero;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* wrow = wb + (size_t)(c4 + e) * g.K;   // wave-uniform: scalar loads
#pragma unroll
      for (int py = 0; py < S; ++py)
#pragma unroll
        for (int px = 0; px < S; ++px)
#pragma unroll
          for (int ay = 0; ay < AY; ++ay)
#pragma unroll
            for (int ax = 0; ax < AY; ++ax) {
              const int ky = py + S * ay, kx = px + S * ax;
              if (ky < KS && kx < KS) {
                const float de = d[ay][ax][e];
                const f32x4 dd = {de, de, de, de};
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                  acc[py * S + px][q] = __builtin_elementwise_fma(dd, *(const f32x4u*)(wrow + (ky * KS + kx) * (4 * NQ) + 4 * q), acc[py * S + px][q]);
              }
            }
    }
  }
  // Epilogue. Written naively each store instruction would put 16 B per lane at a 16*NC-byte stride: partial
  // cache lines, which the memory side answers with fills and repeated write-backs (PMC: 2.2x the algorithmic
  // WRITE_SIZE, +75 % FETCH_SIZE). The wave's row segment is therefore transposed through a wave-private LDS
  // strip so that lane L of store j handles 16-byte chunk j*64 + L of it: whole lines per instruction, for
  // the ReLU-mask read as well.
  constexpr int NC = S * NQ;                 // 16-byte chunks per thread and image row
  __shared__ f32x4 stg[kThreads / 64][64 * NC];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int total = a.B * Yb * Xb;
  const int wave_base = blockIdx.x * kThreads + wave * 64;
  const float inv_xb = 1.0f / (float)Xb, inv_yb = 1.0f / (float)Yb;
#pragma unroll
  for (int py = 0; py < S; ++py) {
#pragma unroll
    for (int px = 0; px < S; ++px)
#pragma unroll
      for (int q = 0; q < NQ; ++q) stg[wave][lane * NC + px * NQ + q] = acc[py * S + px][q];
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the strip is written
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const int w = j * 64 + lane;
      const int src = w / NC, c = w - src * NC;          // NC is a power of two
      const int sidx = wave_base + src;
      const int sx = sidx - fast_div(sidx, Xb, inv_xb) * Xb;
      const int st1 = fast_div(sidx, Xb, inv_xb);
      const int sb = fast_div(st1, Yb, inv_yb);
      const int sy = st1 - sb * Yb;
      const int pxx = c / NQ, qq = c - pxx * NQ;
      const int y = sy * S + py, x = sx * S + pxx;
      const bool ok = sidx < total && y < g.H && x < g.W;
      const size_t o = (((size_t)(ok ? sb : 0) * g.H + (ok ? y : 0)) * g.W + (ok ? x : 0)) * (4 * NQ) + 4 * qq;
      const f32x4 xv = *(const f32x4u*)(a.x[pi] + o);
      f32x4 r = stg[wave][w];
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = xv[e] > 0.f ? r[e] : 0.f;
      if (ok) *(f32x4u*)(a.dx[pi] + o) = r;
    }
    __builtin_amdgcn_wave_barrier();         // the strip is reused by the next row
  }
}

// ---------------------------------------------------------------------------------------------
// The same data gradient on the matrix cores, wave-autonomous like k_conv_fwd_narrow (no LDS, no column buffer), for the
// 3x3 / stride-2 layer with 16 input channels (round 3: 43.9 -> 24.3 us; with 8 input channels half of every tile's
// columns are idle and the block kernel above wins, 30.9 vs 38.6 us; DSACT_NO_CONV_DX_MFMA=1 = A/B):
//   an input pixel (y, x) = (2 yq + py, 2 xq + px) receives from the taps ky = py + 2 ay, kx = px + 2 ax (< 3), i.e. from
//   the output pixels (yq - ay, xq - ax): per row parity py (grid.z) a tile is 16 consecutive (b, yq, xq) PAIRS of pixels
//   (px = 0 and 1 of the same xq -- written as one contiguous 2 * Cin floats per lane group, whole lines per wave);
//   A operand = the dY rows of the <= 4 neighbours (one dwordx4 per lane, neighbour and 16-channel group; rows outside
//   the output image zeroed after landing), B operand = W[co][tap][ci] fragments held in registers for the whole wave,
//   D lane (i, g) = dX[pixel i][ci = 4g .. 4g+3]: one 16-byte store per pixel of the pair, ReLU mask applied.
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int PY>
__device__ __forceinline__ void conv_dx_mfma_body(const ConvDxArgs& a, int pi) {
  constexpr int NKK = COUT / 16, NAY = PY == 0 ? 2 : 1;
  const ConvGeom& g = a.g;
  const int lane = threadIdx.x & 63, i = lane & 15, gq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (kThreads / 64) + ((int)threadIdx.x >> 6));
  const int n_waves = (int)gridDim.x * (kThreads / 64);
  const int Yc = (g.H - PY + 1) >> 1, Xq = (g.W + 1) >> 1;
  const int M = a.B * Yc * Xq, n_tiles = (M + 15) >> 4;
  if (wave >= n_tiles) return;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  const float inv_xq = 1.0f / (float)Xq, inv_yc = 1.0f / (float)Yc;
  // weight fragments: wq[ay][kx][kk][e] = W[co = 16kk + 4gq + e][ky = PY + 2ay][kx][ci = i] (0 for i >= CIN)
  f32x4 wq[NAY][3][NKK];
  const float* __restrict__ wb = a.w[pi];
#pragma unroll
  for (int ay = 0; ay < NAY; ++ay)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        f32x4 v = zero;
        if (i < CIN) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = wb[(size_t)(16 * kk + 4 * gq + e) * g.K + ((PY + 2 * ay) * 3 + kx) * CIN + i];
        }
        wq[ay][kx][kk] = v;
      }
  const float* __restrict__ dyb = a.dy[pi];
  // (X0 / X1: the forward activations behind the ReLU mask of the tile's two pixels -- requested WITH the tile's operands, one
  //  tile ahead: loaded in the epilogue they cost `s_waitcnt vmcnt(0)` in front of every store, which also drained the next
  //  tile's operand requests -- nothing was in flight under the MFMAs; round 6, scripts/isa_wait_audit.py)
  struct Tile { f32x4 A[NAY][2][NKK]; f32x4 X0, X1; bool v[NAY][2]; int b, yq, xq; bool ok; };
  auto load_tile = [&](int tile, Tile& T) {
    const int m = tile * 16 + i;
    T.ok = m < M;
    const int mc = T.ok ? m : M - 1;
    const int t1 = fast_div(mc, Xq, inv_xq);
    T.xq = mc - t1 * Xq;
    T.b = fast_div(t1, Yc, inv_yc);
    T.yq = t1 - T.b * Yc;
#pragma unroll
    for (int ay = 0; ay < NAY; ++ay)
#pragma unroll
      for (int ax = 0; ax < 2; ++ax) {
        const int oy = T.yq - ay, ox = T.xq - ax;
        T.v[ay][ax] = T.ok && oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
        const float* row = dyb + (((size_t)T.b * g.OH + (T.v[ay][ax] ? oy : 0)) * g.OW + (T.v[ay][ax] ? ox : 0)) * COUT + 4 * gq;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) T.A[ay][ax][kk] = *(const f32x4u*)(row + 16 * kk);   // raw: masked at use
      }
    const size_t o = (((size_t)T.b * g.H + 2 * T.yq + PY) * g.W + 2 * T.xq) * CIN + (4 * gq < CIN ? 4 * gq : 0);
    T.X0 = *(const f32x4u*)(a.x[pi] + o);
    T.X1 = *(const f32x4u*)(a.x[pi] + o + (2 * T.xq + 1 < g.W ? CIN : 0));   // (clamped: an odd image width has no second pixel in its last pair)
  };
  auto compute = [&](const Tile& T) {
    f32x4 acc0 = zero, acc1 = zero;      // px = 0, px = 1
#pragma unroll
    for (int ay = 0; ay < NAY; ++ay)
#pragma unroll
      for (int ax = 0; ax < 2; ++ax)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
          const f32x4 av = T.v[ay][ax] ? T.A[ay][ax][kk] : zero;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[ay][2 * ax][kk][e], av[e], acc0, 0, 0, 0);          // kx = 0 / 2
            if (ax == 0) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[ay][1][kk][e], av[e], acc1, 0, 0, 0);  // kx = 1
          }
        }
    if (T.ok && 4 * gq < CIN) {
      const int y = 2 * T.yq + PY, x0 = 2 * T.xq;
      const size_t o = (((size_t)T.b * g.H + y) * g.W + x0) * CIN + 4 * gq;
      const f32x4 xv0 = T.X0;
      f32x4 r0 = acc0;
#pragma unroll
      for (int e = 0; e < 4; ++e) r0[e] = xv0[e] > 0.f ? r0[e] : 0.f;
      *(f32x4u*)(a.dx[pi] + o) = r0;
      if (x0 + 1 < g.W) {
        const f32x4 xv1 = T.X1;
        f32x4 r1 = acc1;
#pragma unroll
        for (int e = 0; e < 4; ++e) r1[e] = xv1[e] > 0.f ? r1[e] : 0.f;
        *(f32x4u*)(a.dx[pi] + o + CIN) = r1;
      }
    }
  };
  Tile Ta, Tb;
  const int last = n_tiles - 1;
  load_tile(wave, Ta);
  for (int tile = wave; tile < n_tiles; tile += 2 * n_waves) {
    const int nx = tile + n_waves;
    load_tile(nx < last ? nx : last, Tb);     // unconditional (clamped): no branch between the loads and the MFMAs
    __builtin_amdgcn_sched_barrier(0);        // (the requests stay in front of the MFMAs: hipcc sank them behind most of them)
    compute(Ta);
    if (nx >= n_tiles) break;
    const int n2 = nx + n_waves;
    load_tile(n2 < last ? n2 : last, Ta);
    __builtin_amdgcn_sched_barrier(0);
    compute(Tb);
  }
}
template <int CIN, int COUT>
__global__ void __launch_bounds__(kThreads) k_conv_dx_mfma(ConvDxArgs a) {
  if (blockIdx.z == 0) conv_dx_mfma_body<CIN, COUT, 0>(a, (int)blockIdx.y);
  else conv_dx_mfma_body<CIN, COUT, 1>(a, (int)blockIdx.y);
}

// sums the partials in a fixed order, writes the gradient and (single-GPU path) applies Adam / Polyak.
// ONE launch for all layers of all differentiated stacks (it runs after the whole conv backward, so every use of the
// pre-update weights is behind it): six tiny reduces as separate launches cost 65 us of launch latency per update.
struct ConvReduceLayer {
  int Cout, K, K1p, n_chunks, quads;   // quads = Cout * K1p / 4
  int block_begin;                     // first block (grid.x) of this layer
  int wide;                            // 1: 16 quads x 16 chunk lanes per block (many chunks); 0: 256 quads x 1 lane
};
struct ConvReduceProb {
  const float* part;
  long long w_idx, b_idx;   // arena index of this layer's weight / bias block
};
struct ConvReduceArgs {
  ConvReduceLayer L[kMaxConv];
  ConvReduceProb p[kMaxConv][3];
  int n_layers, n_prob;
  FusedOpt fo;
};

#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_conv_dw_reduce(ConvReduceArgs a) {
  __shared__ f32x4 red[kThreads];
  int j = 0;
#pragma unroll
  for (int q = 1; q < kMaxConv; ++q) if (q < a.n_layers && (int)blockIdx.x >= a.L[q].block_begin) j = q;
  const ConvReduceLayer& Ly = a.L[j];
  const ConvReduceProb& t = a.p[j][blockIdx.y];
  const int tid = threadIdx.x;
  const int QL = Ly.wide ? 16 : kThreads, CL = Ly.wide ? 16 : 1;
  const int ql = tid % QL, cl = tid / QL;
  const int q = ((int)blockIdx.x - Ly.block_begin) * QL + ql;
  const bool qv = q < Ly.quads;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (qv) {
    // chunk lane cl sums chunks cl, cl+CL, ...: NU independent loads per trip (8: the ~500 chunks of the first layers were
    // 8 dependent round trips per lane at 4; same summation order)
    constexpr int NU = 8;
    for (int c0 = cl; c0 < Ly.n_chunks; c0 += NU * CL) {
      f32x4 v[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int c = c0 + CL * u;
        v[u] = *(const f32x4u*)(t.part + ((size_t)(c < Ly.n_chunks ? c : 0) * Ly.quads + q) * 4);
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) if (c0 + CL * u < Ly.n_chunks) s += v[u];
    }
  }
  if (Ly.wide) {               // uniform per block
    red[tid] = s;
    __syncthreads();
    if (cl != 0 || !qv) return;
#pragma unroll
    for (int u = 1; u < 16; ++u) s += red[u * 16 + ql];
  } else if (!qv) return;
  const int co = (q * 4) / Ly.K1p, kk = q * 4 - co * Ly.K1p;
  if (kk > Ly.K) return;                               // padding quad
  const FusedOpt& fo = a.fo;
  const bool is_bias = kk == Ly.K;
  const long long oi = is_bias ? t.b_idx + co : t.w_idx + (long long)co * Ly.K + kk;
  const int nel = is_bias ? 1 : 4;
  if (is_bias) fo.grads[oi] = s[0];
  else *(f32x4u*)(fo.grads + oi) = s;
  if (fo.st == nullptr) return;
  const bool is_q = oi < fo.n_q2;
  const bool delayed = fo.st->do_delayed != 0;
  if (!(is_q || delayed)) return;
  const float ss = is_q ? fo.st->ss_q : fo.st->ss_pi, bc2 = is_q ? fo.st->bc2_q : fo.st->bc2_pi;
  if (!is_bias) {   // one 16-byte access per stream instead of four 4-byte ones (same per-element arithmetic)
    f32x4 op = *(const f32x4u*)(fo.online + oi), om = *(const f32x4u*)(fo.adam_m + oi), ov = *(const f32x4u*)(fo.adam_v + oi);
    f32x4 ot = {0.f, 0.f, 0.f, 0.f};
    if (delayed) ot = *(const f32x4u*)(fo.target + oi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float pe = op[e], me = om[e], ve = ov[e];
      adam_update(pe, me, ve, s[e], fo.b1w, fo.beta2, fo.b2w, ss, bc2, fo.eps);
      op[e] = pe; om[e] = me; ov[e] = ve;
      if (delayed) ot[e] = polyak_update(ot[e], pe, fo.polyak, fo.one_minus_polyak);
    }
    *(f32x4u*)(fo.online + oi) = op; *(f32x4u*)(fo.adam_m + oi) = om; *(f32x4u*)(fo.adam_v + oi) = ov;
    if (delayed) *(f32x4u*)(fo.target + oi) = ot;
    return;
  }
  for (int e = 0; e < nel; ++e) {
    float pe = fo.online[oi + e], me = fo.adam_m[oi + e], ve = fo.adam_v[oi + e];
    adam_update(pe, me, ve, s[e], fo.b1w, fo.beta2, fo.b2w, ss, bc2, fo.eps);
    fo.online[oi + e] = pe; fo.adam_m[oi + e] = me; fo.adam_v[oi + e] = ve;
    if (delayed) fo.target[oi + e] = polyak_update(fo.target[oi + e], pe, fo.polyak, fo.one_minus_polyak);
  }
}
#endif

// ---------------------------------------------------------------------------------------------
// col2im: dX[b,y,x,ci] = relu'(x[b,y,x,ci]) * sum over the (ky,kx) whose window covers (y,x) of
//   dCol[(b,(y-ky)/s,(x-kx)/s)][(ky*KS+kx)*Cin + ci]; fixed (ky,kx) order => deterministic.
// One thread per 4 channels of one input pixel.
// ---------------------------------------------------------------------------------------------
struct Col2imArgs {
  ConvGeom g;
  const float* dcol[3];   // [M][K]
  const float* x[3];      // layer input activations (post-ReLU of the previous layer) [B*H*W][Cin]
  float* dx[3];           // [B*H*W][Cin]
  int n_prob;
  int B;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_col2im(Col2imArgs a) {
  const ConvGeom& g = a.g;
  const int pi = blockIdx.y;
  const int c4n = g.Cin >> 2;
  const long long total = (long long)a.B * g.H * g.W * c4n;
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= total) return;
  const int c4 = (int)(e % c4n);
  long long pix = e / c4n;
  const int x = (int)(pix % g.W); pix /= g.W;
  const int y = (int)(pix % g.H);
  const int b = (int)(pix / g.H);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const float* dc = a.dcol[pi];
  for (int ky = 0; ky < g.KS; ++ky) {
    const int ty = y - ky;
    if (ty < 0 || ty % g.stride) continue;
    const int oy = ty / g.stride;
    if (oy >= g.OH) continue;
    for (int kx = 0; kx < g.KS; ++kx) {
      const int tx = x - kx;
      if (tx < 0 || tx % g.stride) continue;
      const int ox = tx / g.stride;
      if (ox >= g.OW) continue;
      const size_t m = ((size_t)b * g.OH + oy) * g.OW + ox;
      s += *(const f32x4u*)(dc + m * g.K + (ky * g.KS + kx) * g.Cin + c4 * 4);
    }
  }
  const size_t o = (size_t)e * 4;
  const f32x4 xv = *(const f32x4u*)(a.x[pi] + o);
#pragma unroll
  for (int q = 0; q < 4; ++q) s[q] = xv[q] > 0.f ? s[q] : 0.f;
  *(f32x4u*)(a.dx[pi] + o) = s;
}
#endif

// ---------------------------------------------------------------------------------------------
// features: last conv activation [B*P][C] (pixel-major) <-> flattened NCHW feature columns of the MLP
// input rows (img.view(B,-1), networks/cnn.py:233,455): X[b][c*P + p]
// ---------------------------------------------------------------------------------------------
struct FeatArgs {
  const float* act[6];    // last-layer activations of the stacks
  float* dst0[6]; float* dst1[6];   // MLP input rows fed by the stack (dst1 may be null)
  int n_stack, B, P, C, ldx;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_feat_scatter(FeatArgs a) {
  const int st = blockIdx.y;
  const int F = a.P * a.C;
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= (long long)a.B * F) return;
  const int b = (int)(e / F), f = (int)(e - (long long)b * F);
  const int c = f / a.P, p = f - c * a.P;
  const float v = a.act[st][((size_t)b * a.P + p) * a.C + c];
  a.dst0[st][(size_t)b * a.ldx + f] = v;
  if (a.dst1[st]) a.dst1[st][(size_t)b * a.ldx + f] = v;
}
#endif
struct FeatBwdArgs {
  const float* dfeat[3];  // [B x F]
  const float* act[3];    // last-layer activations (ReLU mask)
  float* dy[3];           // [B*P][C]
  int n_stack, B, P, C;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_feat_bwd(FeatBwdArgs a) {
  const int st = blockIdx.y;
  const int F = a.P * a.C;
  const long long e = (long long)blockIdx.x * kThreads + threadIdx.x;
  if (e >= (long long)a.B * F) return;
  const int b = (int)(e / F);
  const int r = (int)(e - (long long)b * F);   // (p, c) order: coalesced stores
  const int p = r / a.C, c = r - p * a.C;
  const float g = a.dfeat[st][(size_t)b * F + c * a.P + p];
  a.dy[st][e] = a.act[st][e] > 0.f ? g : 0.f;
}
#endif

// ---------------------------------------------------------------------------------------------
// image minibatch gather (training/replay_buffer.py:85-90 for obsv_dim = (C,H,W)): replay rows hold the
// image as the environment delivers it (C,H,W); the staged copy is pixel-major. Coalesced 16-byte
// stores; the 4 source floats of a store come from <= 4 channel planes of neighbouring pixels.
// ---------------------------------------------------------------------------------------------
struct ImgGatherArgs {
  const float* rb_obs; const float* rb_obs2; const float* rb_act; const float* rb_rew; const float* rb_done;
  const int* idx_table; int idx_rows; int use_dev; int host_row;
  const DevState* st;
  float* img0; float* img2;
  float* Xa0; float* Xa1;   // rows receiving the replayed action (q1 / q2 on (obs, act))
  float* rew; float* done;
  int B, C, HW, A, F, ldx;
  int chunks;               // blocks per image
  // fused step flows: block 0 does the per-step bookkeeping, chunk-0 blocks draw the device noise of their row,
  // spare blocks [B*chunks, ...) refresh the padded first-layer weight copies (as k_gather does for the MLP nets)
  int bookkeeping, advance_counters; long long host_it;
  DevState* stw;
  StepHyper hp; NoiseArgs nz; RepackArgs rp;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_gather_img(ImgGatherArgs a) {
  if ((int)blockIdx.x >= a.B * a.chunks) {
    repack_rows(a.rp, (int)blockIdx.x - a.B * a.chunks, threadIdx.x);
    return;
  }
  const int r = blockIdx.x / a.chunks, ck = blockIdx.x - r * a.chunks;
  const int trow = a.use_dev ? (int)(a.st->seq_next % a.idx_rows) : a.host_row;
  const long long src = a.idx_table[(size_t)trow * a.B + r];
  const size_t O = (size_t)a.C * a.HW;
  const float* so = a.rb_obs + (size_t)src * O;
  const float* so2 = a.rb_obs2 + (size_t)src * O;
  float* d0 = a.img0 + (size_t)r * O;
  float* d2 = a.img2 + (size_t)r * O;
  const int n4 = (int)(O >> 2);   // C*HW % 4 == 0 is checked at create time
  const float inv_c = 1.0f / (float)a.C;
  if (a.C == 3 && (a.HW & 3) == 0) {
    // RGB planes: a thread takes 4 consecutive pixels -- one 16-byte load per plane and image, the 3 x 4 block transposed
    // in registers, three 16-byte stores per image (dword loads of the generic loop below: 4x the load instructions,
    // 33 us for the 115 MB of this launch)
    const int nq = a.HW >> 2;
    for (int q = ck * kThreads + threadIdx.x; q < nq; q += a.chunks * kThreads) {
      f32x4 p[3], p2[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        p[c] = *(const f32x4*)(so + (size_t)c * a.HW + 4 * q);
        p2[c] = *(const f32x4*)(so2 + (size_t)c * a.HW + 4 * q);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {     // output quad j holds elements 4j .. 4j+3 of (pixel e, channel c) = e*3 + c
        f32x4 v, w;
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int o = 4 * j + e4;
          v[e4] = p[o % 3][o / 3];
          w[e4] = p2[o % 3][o / 3];
        }
        *(f32x4*)(d0 + (size_t)q * 12 + 4 * j) = v;
        *(f32x4*)(d2 + (size_t)q * 12 + 4 * j) = w;
      }
    }
  } else
  for (int q = ck * kThreads + threadIdx.x; q < n4; q += a.chunks * kThreads) {
    f32x4 v, w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int o = q * 4 + e;
      const int pix = fast_div(o, a.C, inv_c), c = o - pix * a.C;
      v[e] = so[(size_t)c * a.HW + pix];
      w[e] = so2[(size_t)c * a.HW + pix];
    }
    *(f32x4*)(d0 + (size_t)q * 4) = v;
    *(f32x4*)(d2 + (size_t)q * 4) = w;
  }
  if (ck == 0) {
    const int t = threadIdx.x;
    if (a.rb_act && t < a.A) {
      const float av = a.rb_act[(size_t)src * a.A + t];
      a.Xa0[(size_t)r * a.ldx + a.F + t] = av;
      a.Xa1[(size_t)r * a.ldx + a.F + t] = av;
    }
    if (a.rb_act && t == 0) { a.rew[r] = a.rb_rew[src]; a.done[r] = a.rb_done[src]; }
    if (a.bookkeeping) {
      const long long it = a.use_dev ? a.st->it_next : a.host_it;
      if (a.nz.seed != 0) fill_noise_rows(a.nz, it, trow, r, r + 1, a.A, t, kThreads);
      if (blockIdx.x == 0 && t == 0) prologue_duties(a.stw, it, a.advance_counters, a.hp);
    }
  }
}
#endif

// replay ring write for image rows (wide rows: a block per row)
struct ImgScatterArgs {
  const float* s_obs; const float* s_obs2; float* rb_obs; float* rb_obs2;
  long long ptr, cap; int n; long long O;
};
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(kThreads) k_ring_write_img(ImgScatterArgs a) {
  const int i = blockIdx.y;
  const long long dst = (a.ptr + i) % a.cap;
  const long long n4 = a.O >> 2;
  for (long long q = (long long)blockIdx.x * kThreads + threadIdx.x; q < n4; q += (long long)gridDim.x * kThreads) {
    *(f32x4*)(a.rb_obs + dst * a.O + q * 4) = *(const f32x4*)(a.s_obs + (long long)i * a.O + q * 4);
    *(f32x4*)(a.rb_obs2 + dst * a.O + q * 4) = *(const f32x4*)(a.s_obs2 + (long long)i * a.O + q * 4);
  }
}
#endif

}  // namespace dsact
