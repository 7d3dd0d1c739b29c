// dsact_chain.h -- row-slice fused MLP chains of the DSAC-T update (gfx950, wave64, fp32 MFMA).
//
// Forward layers and the hidden-layer backward  dZ[l-1] = (dZ[l] W_l) * gelu'(z[l-1])  are independent per batch
// row, so ONE workgroup can take a slice of the minibatch through a whole network chain: first layer -> hidden
// layers -> output layer -> tanh-Gaussian rsample / softplus epilogue (forward), or loss -> output-layer backward ->
// hidden layers -> dL/d(action) (backward), with the activations in LDS and no inter-workgroup synchronisation.
// One update becomes 3 launches at batch <= 256 (4 at 512, 5 above; the tile path: 14):
//   k_chain_fwd2     group A: policy(obs), policy_target(obs2), q1/q2(obs,act), + the obs2 part of q1_t/q2_t's first layer;
//                    group B (same launch, per-slice ready flags): q1_t/q2_t(obs2,act2), q1/q2(obs,new_act): first layer =
//                    saved obs part + K=A action part            (batch > 256: two launches, k_chain_fwd A then B)
//   k_chain_bwd_q    DSAC-T loss (dsac_v2.py:218-318) + dZ chains of q1c,q2c,q1p,q2p + dL/d new_act (+ the next update's gather)
//   k_chain_bwd_pi   rsample backward + policy dZ chain   (+ the critics' dW/Adam tiles on the idle CUs; batch <= 512: + the
//                    policy's own dW/Adam(/Polyak) tiles, which wait for the chain's arrival counter, + the block that closes
//                    the update)
//   k_dw2            policy dW/Adam(/Polyak) tiles + close of the update, where they do not ride in k_chain_bwd_pi
// Reference math: networks/mlp.py:79-127, utils/act_distribution_cls.py:44-54, dsac_v2.py:150-318.
//
// Shape of a chain workgroup (measurements: scripts/ubench/{slice_gemm,slice_gemm44,cu_stream,mfma_operands}.hip,
// profiles/r02_ubench_*.txt, profiles/r02_chain_*_timeline.txt):
//   * f32 MFMA and the f32 VALU are the same lanes (a VALU-FMA wave beside an MFMA wave on one SIMD: time = sum), so a
//     CU's floor for a slice is MFMA time + epilogue VALU time; the only lever left is MORE CUs per chain, i.e. FEWER
//     rows per workgroup. v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 rows x 4 outputs, K = 1, 8 cycles) maps lane <->
//     output feature and accumulator register <-> batch row, so a workgroup owns 4*RG rows (RG = 2: 8 rows, 32 slices
//     of a 256 minibatch) and wave w the outputs [64w, 64w+64) -- half the rows and half the per-layer MFMA time of
//     the 16x16x4 formulation this file started with (3.9 + 1.2 us per layer measured in the step -> 2.x + 0.6).
//   * every weight byte is used by exactly ONE wave, so weights go L2 -> registers, never through LDS, and a wave-load
//     must be ONE contiguous KB (a CU streams 64 B/clk that way, 14 B/clk with the row-major 16 B-per-row pattern the
//     operand implies): hence the fragment-major ("packed", style 44) weight copies, kept fresh by the Adam tiles.
//   * loads are issued between MFMA groups (a wave issues in order) and run kPD steps ahead ACROSS layer boundaries:
//     the weight stream never drains at an epilogue or barrier (raw s_barrier: __syncthreads() waits for vmcnt(0));
//     LDS operands are addressed by OFFSET into the LDS array (a generic pointer makes them flat loads, which count
//     on vmcnt), and are read one step ahead.
#pragma once
#include "dsact_kernels.h"

namespace dsact {

#ifndef DSACT_TILE_NAP
#define DSACT_TILE_NAP 24   // s_sleep units (64 cycles) between two polls of an arrival counter (ArriveWait)
#endif
constexpr int kChMaxL = 4;     // == DSACT_MAX_HIDDEN_LAYERS
#ifndef DSACT_KPD
#define DSACT_KPD 16
#endif
constexpr int kPD = DSACT_KPD;        // weight steps (4 k of a 64-output tile = 1 KB per wave-load) in flight per wave; every
                               // stream segment is a multiple of kPD steps (64 k) long

// phase stamps of the chain kernels (instrumented builds, -DDSACT_TIMELINE): [block][16] shader-clock values
#ifdef DSACT_TIMELINE
#define CTL(buf, k) do { if ((buf) && threadIdx.x == 0 && blockIdx.x < 1024) (buf)[blockIdx.x * 16 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
// slots 14 / 15: workgroup begin / end on the chip-wide 100 MHz counter (the cycle counter is per XCD: no skew across them)
#define CTLR(buf, k) do { if ((buf) && threadIdx.x == 0 && blockIdx.x < 1024) (buf)[blockIdx.x * 16 + (k)] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#define CTLV(buf, k, v) do { if ((buf) && threadIdx.x == 0 && blockIdx.x < 1024) (buf)[blockIdx.x * 16 + (k)] = (long long)(v); } while (0)
#else
#define CTL(buf, k) do {} while (0)
#define CTLR(buf, k) do {} while (0)
#define CTLV(buf, k, v) do {} while (0)
#endif


// ---- the weight stream (style-44 tensors: [tile of 64 rows][step][lane][4]) ---------------------------------------
struct WStr { f32x4 b[kPD]; };

// first kPD steps of a stream that starts at step s_lo of the wave's tile `cur` (wave-uniform pointer)
__device__ __forceinline__ void stream_prologue(WStr& ws, const float* cur, int s_lo, int lane4) {
#pragma unroll
  for (int u = 0; u < kPD; ++u) ws.b[u] = gload4(cur + (size_t)(s_lo + u) * 256 + lane4);
}

// acc[g][p] += X[rows 4g..4g+3][4 k of step s] (x) W[64 outputs of the wave][same k] over steps [s_lo, s_hi) of `cur`
// ((s_hi - s_lo) % kPD == 0; two accumulators per row group, k parity p, keep dependent MFMAs 4 instructions apart).
// After a step's MFMAs its buffer is refilled with the step kPD positions further down the stream: from `cur` while it
// lasts, then from `nxt` starting at its step nxt_s0 (has_nxt == false: the stream ends; spare slots re-read a valid
// address). lds + xs: this lane's LDS operand (row lane&3 of group 0; group g is 4g rows further; step s is 4s floats
// further), read one step ahead.
template <int RG>
__device__ __forceinline__ void gemm44_seg(WStr& ws, const float* cur, int s_lo, int s_hi, const float* nxt, int nxt_s0,
                                           bool has_nxt, const float* lds, int xs, int ld, int lane4, f32x4 (&acc)[RG][2]) {
  f32x4 a0[RG], a1[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) a0[g] = *(const f32x4*)(lds + xs + 4 * g * ld + 4 * s_lo);
  for (int s0 = s_lo; s0 < s_hi; s0 += kPD) {
    // segment lengths are multiples of kPD and a refill looks kPD steps ahead, so the source of a whole trip's refills
    // is uniform: the next kPD steps of `cur`, or (last trip) the first kPD steps of what follows
    const float* src = s0 + kPD < s_hi ? cur + (size_t)(s0 + kPD) * 256
                                       : (has_nxt ? nxt + (size_t)nxt_s0 * 256 : cur + (size_t)s_lo * 256);
#pragma unroll
    for (int u = 0; u < kPD; ++u) {
      const int s = s0 + u;
      // operand of the next step (one step past the end at the last step: unused)
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        if (u & 1) a0[g] = *(const f32x4*)(lds + xs + 4 * g * ld + 4 * (s + 1));
        else a1[g] = *(const f32x4*)(lds + xs + 4 * g * ld + 4 * (s + 1));
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int g = 0; g < RG; ++g)
          acc[g][e & 1] = __builtin_amdgcn_mfma_f32_4x4x1f32((u & 1) ? a1[g][e] : a0[g][e], ws.b[u][e], acc[g][e & 1], 0, 0, 0);
        if (e == 1) __builtin_amdgcn_sched_barrier(0);   // keeps MFMAs on one accumulator 2*RG instructions apart
      }
      ws.b[u] = gload4(src + (size_t)u * 256 + lane4);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// ---- narrow products (output layers, dL/d action) on v_mfma_f32_16x16x4_f32, style-16 tensors ---------------------
// NTO <= NTO_MAX output tiles of 16; the contraction (W = 64*NW) is split over the NW waves, 4 chunks of 16 k each;
// every wave leaves its partial tiles in LDS and the caller adds them in wave order. The operand rows are the
// workgroup's R rows (lane i reads row i & (R-1): with R < 16 the upper output rows are duplicates nobody reads).
template <int NTO_MAX>
struct NarrowFrags { f32x4 w[NTO_MAX][4]; };

template <int NTO_MAX>
__device__ __forceinline__ void narrow_load(NarrowFrags<NTO_MAX>& f, const float* wf, int C, int nto, int wave, int lane4) {
#pragma unroll
  for (int t = 0; t < NTO_MAX; ++t) {
    if (t < nto) {   // wave-uniform
#pragma unroll
      for (int c = 0; c < 4; ++c) f.w[t][c] = gload4(wf + ((size_t)t * C + (size_t)wave * 4 + c) * 256 + lane4);
    }
  }
}
template <int NTO_MAX>
__device__ __forceinline__ void narrow_mma(const NarrowFrags<NTO_MAX>& f, int nto, int wave, float* lds, int xs, int red, int lane) {
  f32x4 a[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) a[c] = *(const f32x4*)(lds + xs + 16 * (wave * 4 + c));
#pragma unroll
  for (int t = 0; t < NTO_MAX; ++t) {
    if (t < nto) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[t][c][e], a[c][e], acc, 0, 0, 0);
      *(f32x4*)(lds + red + ((wave * NTO_MAX + t) * 64 + lane) * 4) = acc;
    }
  }
}
// element (row m, output n) of the product: the NW wave partials added in wave order
template <int NTO_MAX, int NW>
__device__ __forceinline__ float narrow_get(const float* lds, int red, int m, int n) {
  const int t = n >> 4, ln = (((n & 15) >> 2) << 4) + m, e = n & 3;
  float s = lds[red + ((0 * NTO_MAX + t) * 64 + ln) * 4 + e];
#pragma unroll
  for (int w = 1; w < NW; ++w) s += lds[red + ((w * NTO_MAX + t) * 64 + ln) * 4 + e];
  return s;
}

// sum over the TPR consecutive lanes that share a batch row in the row phases (TPR = 4 .. 64, power of two)
template <int TPR>
__device__ __forceinline__ float rowN_sum(float v) {
  v += dpp_mov<0xB1>(v);                          // pairs
  v += dpp_mov<0x4E>(v);                          // quads
  if (TPR >= 8) v += dpp_mov<0x141>(v);           // row_half_mirror: 8
  if (TPR >= 16) v += dpp_mov<0x140>(v);          // row_mirror: 16
  if (TPR >= 32) { float a, b; swap16(v, a, b); v = a + b; }
  if (TPR >= 64) { float a, b; swap32(v, a, b); v = a + b; }
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
// The same placement as a table, for launches whose units differ in rows per workgroup (hence in slice count): the block
// on XCD x = b & 7 in dispatch round b >> 3 runs slice base[x] + stride[x] * round of unit[x] (unit < 0: XCD idle).
struct XcdMap { int unit[8], base[8], stride[8]; };
inline XcdMap xcd_map_uniform(int n_units) {
  XcdMap m;
  const int rep = n_units >= 5 ? 1 : (n_units >= 3 ? 2 : (n_units == 2 ? 4 : 8));
  for (int x = 0; x < 8; ++x) {
    const bool on = x < n_units * rep;
    m.unit[x] = on ? x % n_units : -1; m.base[x] = on ? x / n_units : 0; m.stride[x] = rep;
  }
  return m;
}
// unit u gets share[u] consecutive XCDs (sum of shares <= 8), its slices dealt round-robin over them
inline XcdMap xcd_map_shares(int n_units, const int* share) {
  XcdMap m;
  int x = 0;
  for (int u = 0; u < n_units; ++u)
    for (int k = 0; k < share[u] && x < 8; ++k, ++x) { m.unit[x] = u; m.base[x] = k; m.stride[x] = share[u]; }
  for (; x < 8; ++x) { m.unit[x] = -1; m.base[x] = 0; m.stride[x] = 1; }
  return m;
}

// LDS carve-up shared by the chain kernels (floats)
struct ChainLds {
  int ld_in, ld_h, off_in, off_h0, off_h1, off_red, off_sc, off_carry, total;
};
__host__ __device__ inline ChainLds chain_lds(int k_in /*floats of the widest staged operand row*/, int W, int R) {
  ChainLds s;
  s.ld_in = k_in + 8; s.ld_h = W + 8;
  s.off_in = 0;
  s.off_h0 = s.off_in + R * s.ld_in + 16;
  s.off_h1 = s.off_h0 + R * s.ld_h + 16;
  s.off_red = s.off_h1 + R * s.ld_h + 16;
  s.off_sc = s.off_red + (W / 64) * 4 * 64 * 4;   // red: [NW waves][<= 4 tiles][64][4]
  s.off_carry = s.off_sc + 64;              // twin trunks: [R][64] output-layer partial sums of the first trunk
  s.total = s.off_carry + R * 64;
  return s;
}

// agent-scope (sc1) scalar accesses: through to memory / past the non-coherent cache levels (in-launch hand-overs)
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---------------------------------------------------------------------------------------------------------------
// dw2: weight / bias gradients + fused Adam / Polyak from TRANSPOSED fragment-major operands
//   dW[m][n] = sum_b dZ[b][m] X[b][n]: the chain kernels leave dZ^T, H^T, X0^T as style-16 packs [feature][batch]
//   (16-byte stores from their lane = feature layout), so a tile streams its MFMA operands L2 -> registers: no LDS
//   staging, no transposition, tile descriptors in the kernel arguments. A 256-thread workgroup owns a 32x32 block;
//   the batch contraction of the tile (<= 16 chunks of 16 rows) is split over the 4 waves, each multiplying the whole
//   block (2x2 MFMA tiles, 4 KB of operands per 16 MFMAs = 32 B/clk per CU); the 4 partial blocks meet in LDS and
//   wave w finishes 16x16 block w (Adam, Polyak, packed weight mirrors exactly as the tile path's epilogue).
//   Tiles of the first column (n0 == 0) also reduce the bias gradient from the dZ^T fragments they hold.
// ---------------------------------------------------------------------------------------------------------------
struct DwProb {
  const float* At; const float* Xt;   // style-16 packs, C chunks per row tile: [M_pad x B] (output features), [N_pad x B] (inputs)
  long long w_idx, b_idx;             // arena index of W[0][0] and bias[0] (gradient, parameter, moment and target arenas mirror each other)
  int M, N;
  int tiles_n, tile_end;              // 32-wide column tiles; exclusive end of this problem's tile range
  const MirrorDesc* mir;
  int msplit, nsplit;                 // nsplit > 0: the block-diagonal output layer [[w_mean, 0], [0, w_log_std]] of a twin-trunk net
                                      // (networks/cnn.py:224-229): element (m, n) is structurally zero when (m < msplit) != (n < nsplit)
};
constexpr int kMaxDwProb = 3 * 2 * kChMaxL;   // twin trunks: first layer + 2 x hidden layers + output layer per net
struct Dw2Args {
  DwProb p[kMaxDwProb]; int n_prob;
  int tile_ends[kMaxDwProb];   // p[q].tile_end again, contiguous: a tile finds its problem from two wide scalar loads instead
                               // of one strided load per problem
  int C, ct;              // chunks (16 batch rows) per operand row tile; chunks per tile (<= 16 per round, rounds as needed)
  int n_base;             // tiles of one batch range (split-K: tile index = range * n_base + base tile)
  float* gout;            // gradient destination arena (grads; split-K: partial arena 0)
  long long part_stride;  // split-K: floats between the partial arenas
  FusedOpt fo;            // fo.st == nullptr: plain gradient store
  int store_g;            // 0: fused graph replays -- nothing reads the gradient arena, skip its 4.6 MB of stores
};
constexpr int kDw2LdsFloats = 4 * 4 * 64 * 4 + 4 * 2 * 64;
constexpr int dw2_lds_floats(int nwv) { return nwv * 4 * 64 * 4 + nwv * 2 * 64; }   // NWV waves per tile (4: the above)

// `wait`: called between the loads of everything the tile needs from EARLIER launches (the X-side fragments, the Adam /
// Polyak operands) and the first load of the dZ-side fragments -- the policy's tiles of the merged policy-backward
// launch wait there for the policy chain's arrival counter with their other operands already in flight.
struct NoWait { __device__ __forceinline__ void operator()() const {} };
// NWV = waves per tile: the contraction of a round (4 * NWV chunks) is split over them; NWV = 8 (k_dw2 as its own launch
// at batch >= 1024, one tile per CU: two waves per SIMD instead of one hide each other's operand waits) -- the first four
// waves finish the tile as before, summing 8 partial blocks instead of 4 (fixed order).
template <int NSET = 2, typename Wait = NoWait, int NWV = 4>
__device__ __forceinline__ void dw2_tile(const Dw2Args& a, int t, float* lds, Wait wait = Wait()) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4, lane4 = lane * 4;
  const int range = t / a.n_base, bt = t - range * a.n_base;
  int pi = 0;
#pragma unroll
  for (int q = 0; q + 1 < kMaxDwProb; ++q)
    if (q + 1 < a.n_prob && bt >= a.tile_ends[q]) pi = q + 1;
  const DwProb& P = a.p[pi];
  const int local = bt - (pi ? a.p[pi - 1].tile_end : 0);
  const int mt = local / P.tiles_n, nt = local - mt * P.tiles_n;
  const int m0 = 32 * mt, n0 = 32 * nt;
  // ---- the contraction runs in rounds of <= 16 chunks (256 batch rows); in a round the 4 waves split the chunks (wave w:
  //      chunks cb + w*cw + q, q < cw <= 4). The fragments travel in NSET register sets of two chunks each (half rounds):
  //      all are loaded up front -- batch <= 256 is exactly two of them, one round -- and a set is refilled with the half
  //      round NSET ahead as soon as its MFMAs are issued, so longer contractions (batch 512 .. 1024 per range) stream.
  //      NSET = 2 keeps a load 32 MFMAs (~0.4 us) ahead of its use; NSET = 4 (128 fragment registers, 96 MFMAs ahead)
  //      bought nothing at batch 512 / 1024 -- a wave's 16 chunks x 16 MFMAs are the time. Same accumulation order.
  constexpr int RC = 4 * NWV;           // chunks per round
  constexpr int PB = NWV * 4 * 64 * 4;  // floats of the partial blocks in LDS; the bias row sums follow
  const int c_lo = range * a.ct;
  const int n_half = 2 * ((a.ct + RC - 1) / RC);
  f32x4 fa[NSET][2][2], fx[NSET][2][2];
  auto half_ok = [&](int hr, int q2, int& c) {
    const int cb = (hr >> 1) * RC;
    const int ctr = a.ct - cb < RC ? a.ct - cb : RC, cw = (ctr + NWV - 1) / NWV;
    const int q = 2 * (hr & 1) + q2;
    const bool ok = q < cw && wave * cw + q < ctr;
    c = c_lo + cb + (ok ? wave * cw + q : 0);
    return ok;
  };
  auto load_half_x = [&](int set, int hr) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      int c;
      (void)half_ok(hr, q2, c);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) fx[set][q2][b2] = gload4(P.Xt + ((size_t)(2 * nt + b2) * a.C + c) * 256 + lane4);
    }
  };
  auto load_half_a = [&](int set, int hr) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      int c;
      (void)half_ok(hr, q2, c);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) fa[set][q2][b2] = gload4(P.At + ((size_t)(2 * mt + b2) * a.C + c) * 256 + lane4);
    }
  };
  auto load_half = [&](int set, int hr) { load_half_a(set, hr); load_half_x(set, hr); };
#pragma unroll
  for (int st = 0; st < NSET; ++st)
    if (st < n_half) load_half_x(st, st);
  // ---- this lane's share of the epilogue: 16x16 block (wave>>1, wave&1), rows m, columns n .. n+3
  const bool fin = wave < 4;            // the waves that finish the tile (NWV = 8: the other four only multiply)
  const int m = m0 + 16 * ((wave & 3) >> 1) + i, n = n0 + 16 * (wave & 1) + 4 * g;
  const bool in_range = fin && m < P.M && n < P.N, full = n + 3 < P.N;
  const bool fused = a.fo.st != nullptr;
  const long long oi = P.w_idx + (long long)m * P.N + n;
  bool o_delayed = false, o_upd = false;
  float o_ss = 0.f, o_bc2 = 1.f;
  f32x4 op = {0.f, 0.f, 0.f, 0.f}, om = op, ov = op, ot = op;
  if (fused) {
    const bool is_q = P.w_idx < a.fo.n_q2;
    o_delayed = a.fo.st->do_delayed != 0;
    o_ss = is_q ? a.fo.st->ss_q : a.fo.st->ss_pi;
    o_bc2 = is_q ? a.fo.st->bc2_q : a.fo.st->bc2_pi;
    o_upd = is_q || o_delayed;
    if (o_upd && in_range && full) {
      op = *(const f32x4u*)(a.fo.online + oi); om = *(const f32x4u*)(a.fo.adam_m + oi); ov = *(const f32x4u*)(a.fo.adam_v + oi);
      if (o_delayed) ot = *(const f32x4u*)(a.fo.target + oi);
    }
  }
  wait();
#pragma unroll
  for (int st = 0; st < NSET; ++st)
    if (st < n_half) load_half_a(st, st);
  // ---- MFMA: D[row = input feature 4g+reg][col = output feature i]
  f32x4 acc[2][2];
#pragma unroll
  for (int bm = 0; bm < 2; ++bm)
#pragma unroll
    for (int bn = 0; bn < 2; ++bn) acc[bm][bn] = f32x4{0.f, 0.f, 0.f, 0.f};
  float sb[2] = {0.f, 0.f};
  auto mma_half = [&](int set, int hr) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      int c;
      if (half_ok(hr, q2, c)) {   // wave-uniform
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int bm = 0; bm < 2; ++bm)
#pragma unroll
            for (int bn = 0; bn < 2; ++bn)
              acc[bm][bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[set][q2][bn][e], fa[set][q2][bm][e], acc[bm][bn], 0, 0, 0);
#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
          sb[bm] += (fa[set][q2][bm][0] + fa[set][q2][bm][1]) + (fa[set][q2][bm][2] + fa[set][q2][bm][3]);
      }
    }
  };
  // Long contractions (batch >= 512 per range): every half round but the last NSET belongs to a FULL round (each wave has
  // its chunks), so the streaming part runs without a branch -- MFMAs of a set, then its refill, in a fixed order, the
  // first trip peeled: hipcc derives its s_waitcnt counts from the request order it can see on a loop's entry and back
  // edges, and the round-4 form (refill under `if`, per-chunk `if` around the MFMAs) got `vmcnt(0)` before every half --
  // each half waited for the refill requested just before it (one L2 round trip per 32 MFMAs). The last NSET halves (all
  // of a batch <= 256 contraction) keep the wave-uniform checks of a ragged round. Same accumulation order.
  auto mma_half_full = [&](int set) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int bm = 0; bm < 2; ++bm)
#pragma unroll
          for (int bn = 0; bn < 2; ++bn)
            acc[bm][bn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fx[set][q2][bn][e], fa[set][q2][bm][e], acc[bm][bn], 0, 0, 0);
#pragma unroll
      for (int bm = 0; bm < 2; ++bm)
        sb[bm] += (fa[set][q2][bm][0] + fa[set][q2][bm][1]) + (fa[set][q2][bm][2] + fa[set][q2][bm][3]);
    }
  };
  auto stream_trip = [&](int hr) {
#pragma unroll
    for (int st = 0; st < NSET; ++st) {
      mma_half_full(st);
      const int nx = hr + st + NSET;
      load_half(st, nx < n_half ? nx : n_half - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto last_halves = [&](int hr) {
#pragma unroll
    for (int st = 0; st < NSET; ++st)
      if (hr + st < n_half) mma_half(st, hr + st);
  };
  if (NSET < n_half) {
    stream_trip(0);
    int hr = NSET;
    for (; hr + NSET < n_half; hr += NSET) stream_trip(hr);
    last_halves(hr);
  } else {
    last_halves(0);
  }
  // ---- partial blocks -> LDS -> block `wave`
  const bool bias = nt == 0 && P.b_idx >= 0;
#pragma unroll
  for (int bm = 0; bm < 2; ++bm)
#pragma unroll
    for (int bn = 0; bn < 2; ++bn) *(f32x4*)(lds + ((wave * 4 + 2 * bm + bn) * 64 + lane) * 4) = acc[bm][bn];
  if (bias) { lds[PB + (wave * 2 + 0) * 64 + lane] = sb[0]; lds[PB + (wave * 2 + 1) * 64 + lane] = sb[1]; }
  lds_barrier();
  if (NWV > 4 && !fin) return;          // (after the barrier; nothing below synchronises the workgroup)
  f32x4 v = *(const f32x4*)(lds + ((0 * 4 + wave) * 64 + lane) * 4);
#pragma unroll
  for (int w = 1; w < NWV; ++w) v += *(const f32x4*)(lds + ((w * 4 + wave) * 64 + lane) * 4);
  if (P.nsplit > 0 && ((m < P.msplit) != (n < P.nsplit))) v = f32x4{0.f, 0.f, 0.f, 0.f};   // (n, nsplit: multiples of 4)
  float* C = a.gout + range * a.part_stride;
  if (in_range) {
    float* c0 = C + oi;
    if (a.store_g) {
      if (full) *(f32x4u*)c0 = v;
      else for (int e = 0; e < 4 && n + e < P.N; ++e) c0[e] = v[e];
    }
    if (fused && o_upd) {
      if (full) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float pe = op[e], me = om[e], ve = ov[e];
          adam_update(pe, me, ve, v[e], a.fo.b1w, a.fo.beta2, a.fo.b2w, o_ss, o_bc2, a.fo.eps);
          op[e] = pe; om[e] = me; ov[e] = ve;
          if (o_delayed) ot[e] = polyak_update(ot[e], pe, a.fo.polyak, a.fo.one_minus_polyak);
        }
        *(f32x4u*)(a.fo.online + oi) = op; *(f32x4u*)(a.fo.adam_m + oi) = om; *(f32x4u*)(a.fo.adam_v + oi) = ov;
        if (o_delayed) *(f32x4u*)(a.fo.target + oi) = ot;
      } else {
        for (int e = 0; e < 4 && n + e < P.N; ++e) {
          float pe = a.fo.online[oi + e], me = a.fo.adam_m[oi + e], ve = a.fo.adam_v[oi + e];
          adam_update(pe, me, ve, v[e], a.fo.b1w, a.fo.beta2, a.fo.b2w, o_ss, o_bc2, a.fo.eps);
          a.fo.online[oi + e] = pe; a.fo.adam_m[oi + e] = me; a.fo.adam_v[oi + e] = ve;
          op[e] = pe;
          if (o_delayed) { ot[e] = polyak_update(a.fo.target[oi + e], pe, a.fo.polyak, a.fo.one_minus_polyak); a.fo.target[oi + e] = ot[e]; }
        }
      }
    }
  }
  // the packed copies; every lane takes part (the transposed copy is assembled across lane quads with DPP)
  if (fused && o_upd && P.mir) {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    mirror_store4_quad(*P.mir, m, n, P.N, P.M, in_range, in_range ? op : zero, o_delayed, ot, lane);
  }
  // ---- bias gradient of rows m0 .. m0+31: sum of the 4 waves' x 4 lane groups' row sums
  if (bias && tid < 32) {
    const int mb = m0 + tid;
    if (mb < P.M) {
      float sbias = 0.f;
#pragma unroll
      for (int w = 0; w < NWV; ++w)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) sbias += lds[PB + (w * 2 + (tid >> 4)) * 64 + gg * 16 + (tid & 15)];
      const long long bi = P.b_idx + mb;
      if (a.store_g) C[bi] = sbias;
      if (fused && o_upd) {
        float pe = a.fo.online[bi], me = a.fo.adam_m[bi], ve = a.fo.adam_v[bi];
        adam_update(pe, me, ve, sbias, a.fo.b1w, a.fo.beta2, a.fo.b2w, o_ss, o_bc2, a.fo.eps);
        a.fo.online[bi] = pe; a.fo.adam_m[bi] = me; a.fo.adam_v[bi] = ve;
        if (o_delayed) a.fo.target[bi] = polyak_update(a.fo.target[bi], pe, a.fo.polyak, a.fo.one_minus_polyak);
      }
    }
  }
}

// Tile order is (layer, row block, column block): neighbours share operands. The dispatcher deals consecutive block
// ids round-robin over the 8 XCDs, so handing out tiles in block order makes every L2 fetch every layer's operands
// (~6 MB x 8 over the fabric per update). Instead XCD x takes the x-th contiguous eighth of the tile list: an L2 sees
// the operands of one or two layers only. b: block index of a launch whose first tile block sits on XCD 0
// (grid = xcd_chunk_grid(n) blocks); false: padding block.
__host__ __device__ inline int xcd_chunk_grid(int n) { return 8 * ((n + 7) >> 3); }
__device__ __forceinline__ bool xcd_chunk(int b, int n, int& t) {
  const int per = (n + 7) >> 3;
  t = (b & 7) * per + (b >> 3);
  return t < n && (b >> 3) < per;
}

// grid (xcd_chunk_grid(n_tiles) [+ 1], batch ranges): base tiles tile0 .. tile0 + n_tiles - 1 of every range; the
// extra block closes the update
struct Dw2Launch { Dw2Args a; int tile0, n_tiles, finalize; };
template <int NSET, int NWV = 4>
__global__ void __launch_bounds__(64 * NWV) k_dw2(Dw2Launch L) {
  __shared__ __attribute__((aligned(16))) float lds[dw2_lds_floats(NWV)];
  if ((int)blockIdx.x >= xcd_chunk_grid(L.n_tiles)) {
    if (L.finalize && blockIdx.y == 0 && threadIdx.x == 0) finalize_update(L.a.fo);
    return;
  }
  int t;
  if (!xcd_chunk((int)blockIdx.x, L.n_tiles, t)) return;
  dw2_tile<NSET, NoWait, NWV>(L.a, (int)blockIdx.y * L.a.n_base + L.tile0 + t, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// k_adam_pack: the optimiser of the data-parallel graph. After the all-reduce the gradient arena holds the averaged
// gradient; one pass applies Adam (+ Polyak on delayed-update steps) to every weight tensor AND writes its packed
// copies (what the fused weight-gradient tiles do on the single-GPU path) -- instead of a streaming k_adam followed by a
// k_pack that reads the arenas again. Same per-element arithmetic as dw2_tile's epilogue / k_adam (tested bitwise).
// Block = 16 rows x 64 columns of one tensor (~690 blocks at Humanoid 3x256: 16 x 256 blocks were 172, fewer than the CUs);
// lanes are laid out like pack_block (a lane quad = 4 consecutive rows).
// ---------------------------------------------------------------------------------------------------------------
struct AdamPackJob {
  long long w_idx, b_idx;   // arena indices of the weight matrix [N x K] and its bias [N]
  int N, K, col_chunks;     // col_chunks = ceil(K / 64)
  const MirrorDesc* mir;
  int block_end;            // exclusive end of this job's block range (row blocks x col_chunks)
};
struct AdamPackArgs { const AdamPackJob* jobs; int n_jobs, n_blocks; FusedOpt fo; };
#ifndef DSACT_FAMILY_UNIT   // plain kernel: compiled in dsact_api.hip only (dsact_tu.h)
__global__ void __launch_bounds__(256) k_adam_pack(AdamPackArgs a) {
  const int b = (int)blockIdx.x, tid = threadIdx.x;
  if (b >= a.n_blocks) {
    if (tid == 0) finalize_update(a.fo);
    return;
  }
  int ji = 0;
  for (int q = 0; q + 1 < a.n_jobs; ++q) if (b >= a.jobs[q].block_end) ji = q + 1;
  const AdamPackJob J = a.jobs[ji];
  const int local = b - (ji ? a.jobs[ji - 1].block_end : 0);
  const int n0 = (local / J.col_chunks) * 16, k_lo = (local % J.col_chunks) * 64;
  const FusedOpt& fo = a.fo;
  const bool o_delayed = fo.st->do_delayed != 0;
  const bool is_q = J.w_idx < fo.n_q2;
  if (!(is_q || o_delayed)) return;   // the policy is left alone on the off iterations of the delayed update
  const float o_ss = is_q ? fo.st->ss_q : fo.st->ss_pi, o_bc2 = is_q ? fo.st->bc2_q : fo.st->bc2_pi;
  {
    const int e = tid;
    const int n = n0 + (e & 15), k = k_lo + (e >> 4) * 4;
    const bool valid = n < J.N && k < J.K;
    f32x4 op = {0.f, 0.f, 0.f, 0.f}, ot = op;
    if (valid) {
      const long long oi = J.w_idx + (long long)n * J.K + k;
      if (k + 3 < J.K) {
        const f32x4 g = *(const f32x4u*)(fo.grads + oi);
        f32x4 om = *(const f32x4u*)(fo.adam_m + oi), ov = *(const f32x4u*)(fo.adam_v + oi);
        op = *(const f32x4u*)(fo.online + oi);
        if (o_delayed) ot = *(const f32x4u*)(fo.target + oi);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float pe = op[c], me = om[c], ve = ov[c];
          adam_update(pe, me, ve, g[c], fo.b1w, fo.beta2, fo.b2w, o_ss, o_bc2, fo.eps);
          op[c] = pe; om[c] = me; ov[c] = ve;
          if (o_delayed) ot[c] = polyak_update(ot[c], pe, fo.polyak, fo.one_minus_polyak);
        }
        *(f32x4u*)(fo.online + oi) = op; *(f32x4u*)(fo.adam_m + oi) = om; *(f32x4u*)(fo.adam_v + oi) = ov;
        if (o_delayed) *(f32x4u*)(fo.target + oi) = ot;
      } else {
        for (int c = 0; c < 4 && k + c < J.K; ++c) {
          float pe = fo.online[oi + c], me = fo.adam_m[oi + c], ve = fo.adam_v[oi + c];
          adam_update(pe, me, ve, fo.grads[oi + c], fo.b1w, fo.beta2, fo.b2w, o_ss, o_bc2, fo.eps);
          fo.online[oi + c] = pe; fo.adam_m[oi + c] = me; fo.adam_v[oi + c] = ve;
          op[c] = pe;
          if (o_delayed) { ot[c] = polyak_update(fo.target[oi + c], pe, fo.polyak, fo.one_minus_polyak); fo.target[oi + c] = ot[c]; }
        }
      }
    }
    // k stays real for a lane whose ROW is past the end: it still owns row k + lane%4 of the transposed block
    if (J.mir) mirror_store4_quad(*J.mir, n, k < J.K ? k : J.K, J.K, J.N, valid, op, o_delayed, ot, tid);
  }
  if (k_lo == 0 && tid < 16 && n0 + tid < J.N) {
    const long long bi = J.b_idx + n0 + tid;
    float pe = fo.online[bi], me = fo.adam_m[bi], ve = fo.adam_v[bi];
    adam_update(pe, me, ve, fo.grads[bi], fo.b1w, fo.beta2, fo.b2w, o_ss, o_bc2, fo.eps);
    fo.online[bi] = pe; fo.adam_m[bi] = me; fo.adam_v[bi] = ve;
    if (o_delayed) fo.target[bi] = polyak_update(fo.target[bi], pe, fo.polyak, fo.one_minus_polyak);
  }
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// k_chain_fwd
// ---------------------------------------------------------------------------------------------------------------
// SEG_FULL_SPLIT: observation segment, then the two accumulators are merged exactly as SEG_OBS_ONLY's zsave /
// SEG_ACT_FROM_SAVED's zinit would have carried them through memory, then the action segment -- ONE workgroup, the same
// bits as the two-unit form (the pipelined graph's q_target units: no obs-only producer, no saved accumulators)
enum : int { SEG_FULL = 0, SEG_OBS_ONLY = 1, SEG_ACT_FROM_SAVED = 2, SEG_FULL_SAVE = 3, SEG_FULL_SPLIT = 4 };
enum : int { HEAD_NONE = 0, HEAD_POLICY = 1, HEAD_Q = 2 };
// FwdUnit::head = HEAD_* | twin role | (chunks of 16 k per 16-row tile of the output-layer pack << 8; 0: W / 16).
// Twin trunks (the CNN nets' `mean` / `log_std` MLPs over one feature row, networks/cnn.py:214-240,437-461): the output layer is
// the dense [n_out x 2H] matrix [[w_mean, 0], [0, w_log_std]], each trunk multiplies ITS 2H-columns half of it (the structural
// zeros contribute exact +0), and the two partial products add up to the net's outputs. One workgroup runs the two trunk
// units of its slice back to back (k_chain_fwdt): HEAD_TWIN_FIRST leaves its partial outputs in LDS and returns,
// HEAD_TWIN_SECOND adds them to its own before the head's row phase. Or -- the two trunks as workgroups of their own,
// twice the workgroups at half the chain length each: the first trunk hands its partial outputs over through memory
// (FwdUnit::qout = [B][64] partials, agent-scope stores, done[slice]); the second one (late_wait & HW_HEAD: wait0 / zinit
// name the partner's flags / partials) waits for them just before its row phase. The first trunk's blocks come earlier in
// the dispatch order than their partners, so the bounded spin cannot deadlock.
enum : int { HEAD_KIND = 15, HEAD_TWIN_FIRST = 16, HEAD_TWIN_SECOND = 32 };
__host__ __device__ inline int head_code(int kind, int twin_role, int c_out) { return kind | twin_role | (c_out << 8); }
// Round 6 -- OUTPUT activations on the chains (value_output_activation / policy_output_activation, networks/mlp.py:15-20; until now
// the tile-stage kernels only): bits 16-18 of FwdUnit::head = the ACT_* id applied to the head's outputs (0: linear), bit 19 =
// the policy's log-std outputs are NOT activated (policy_std_type "parameter": log_std is a plain parameter, networks/mlp.py:92-97).
// The heads store POST-activation outputs, as the tile path's do (k_heads); the derivative is expressed through them
// (out_act_grad_y) in the backward row phases. Compiled only into the generic-activation instantiations (GA): every shipped
// example is linear, and the default kernels stay exactly as they were.
enum : int { HEAD_OUT_ACT_SHIFT = 16, HEAD_STD_PLAIN = 1 << 19 };
__host__ __device__ inline int head_out_act(int head) { return (head >> HEAD_OUT_ACT_SHIFT) & 7; }
enum : int { HW_LATE = 1, HW_PAIRS_OUT = 2, HW_PAIRS_IN = 4, HW_HEAD = 8 };

struct FwdUnit {
  const float* wf[kChMaxL + 1];     // packed weights per layer: style 44 for l < L, style 16 for the output layer (index L)
  const float* bias[kChMaxL + 1];
  const float* x;                   // input rows [B][ldx]: observation part at column 0, action part at column F
  int seg;                          // SEG_*
  int s_act;                        // steps of this net's action segment (0: policy nets)
  float* zsave; const float* zinit; // [B][W] first-layer accumulators after the observation part
  float* H[kChMaxL]; float* G[kChMaxL];   // gelu(z), gelu'(z) per layer, TRANSPOSED style-16 packs [feature][batch]; nullptr: not kept
  float* x0t;                       // transposed pack of the staged input rows [F+A][batch] (one unit writes it), or nullptr
  int head;                         // HEAD_*
  float* logits; float* logp; const float* eps;   // policy head: [B][2A], [B], [B][A]
  float* xact; float* xact2;        // policy head: rows whose action columns (at F) receive the sampled action
  float* qout; float* qstd;         // q head: raw (mean, pre-softplus std) [B][2]; (std, d std/d raw) [B][2] or nullptr
                                    // (a HEAD_TWIN_FIRST unit has no head of its own: qout != nullptr = the [B][64] buffer its partial
                                    //  outputs go to when the partner trunk is another workgroup -- whose zinit reads it, late_wait & HW_HEAD)
  float* part_heads;                // policy head: [slices][2] sums of tanh(mu), sigma; nullptr: none
  // merged A+B launch (k_chain_fwd2): a producer raises done[slice] when everything it wrote is visible chip-wide; a
  // consumer waits for wait0/wait1[its first row / wait_rows] before it reads what the producers wrote. nullptr: no flags
  int* done; const int* wait0; const int* wait1; int wait_rows0, wait_rows1;
  int* zdone;                       // SEG_FULL_SAVE producers: raised as soon as zsave is written (the consumers of the saved
                                    // observation part do not wait for the rest of this unit's chain)
  short rg, act;                    // rows per workgroup / 4 of THIS unit (units of one launch may differ); hidden activation
                                    // of its net (ACT_*, dsact_math.h). (shorts: two FwdArgs must fit the 4 KB of kernel arguments)
  // (late_wait sits at a DWORD-ALIGNED offset, in front of n_slices: a 16-bit field at +2 of a dword of a unit table in device
  //  memory -- k_chain_fwdp / fwdpb / fwdt -- is no scalar load but `global_load_ushort` + `s_waitcnt vmcnt(0)`: the wait drained
  //  the weight stream's prologue and the warm-up touches BEFORE the input rows were requested, one memory round trip of every
  //  chain's start-up; an aligned 16-bit field is widened to s_load_dword. Round 6, ISA.)
  short late_wait;                  // bit 0 (HW_LATE): wait for the producers AFTER the observation segment (only the action columns
                                    // are handed over): the wait hides under this unit's own first 3/4 of a layer.
                                    // Tagged hand-over (pipelined launches): the sampled actions travel as (value, tag) pairs, one
                                    // 8-byte agent-scope store / load each -- the data IS the flag (no store-acknowledge barrier
                                    // + flag on the producer side, no second round trip on the consumer side). bit 1 (HW_PAIRS_OUT,
                                    // policy heads): xact2 is the pair buffer [B][32]; bit 2 (HW_PAIRS_IN): wait0 is that buffer
  short n_slices;                   // slices of this unit
};
constexpr int kMaxFwdUnits = 6;
struct FwdArgs {
  FwdUnit u[kMaxFwdUnits];
  int n_units;
  XcdMap map;                       // block -> (unit, slice)
  int B, F, A, L, ldx;
  int Cb;                           // B / 16: chunks per row tile of the transposed packs
  int s_obs, s_act;                 // steps of the first layer's observation segment / widest action segment (multiples of kPD)
  int v1_stats;
  const float* act_scale; const float* act_center; float lo_ls, hi_ls;
  long long* timeline;
  int* spin_timeout;                // merged launch: set to 1 by a consumer that gave up waiting. The word lives in mapped
                                    // HOST memory: every entry point of the library checks it and fails the call
  int debug_withhold;               // tests only (dsact_debug_set "withhold_flag"): unit 0 / slice 0 never raises its flag
  const long long* tagp;            // tagged hand-over: DevState::tag_seq (advances with every closed update, never reset): tag = low word + 1
  int tpad;                         // steps of padding behind every 64-row tile of the forward packs (experiments: DSACT_PK_PAD)
  int x0_lds;                       // throughput-regime forward (dsact_fat.h): the slice's input rows are staged in LDS (round 6)
};

// Data handed from a producer to a consumer INSIDE the merged launch (sampled actions, saved first-layer accumulators)
// is written and read with agent-scope accesses (sc1: through to memory / past the non-coherent cache levels), so
// neither side needs an L2-wide write-back or invalidate -- a release fence per producer costs an XCD-wide buffer_wbl2
// (measured: the merged launch ran 46 us with fences, 31 us as two launches).

// every outstanding vector-memory operation of this wave (stores included) has been acknowledged, then the workgroup
// barrier: what the producers of an in-launch hand-over run before they raise a flag / bump a counter
__device__ __forceinline__ void stores_acked_barrier() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// producer side of the merged launch: every wave waits until its agent-scope (write-through) stores have been
// acknowledged, the workgroup meets, then one thread raises the flag at agent scope. The wait is EXPLICIT: on gfx950
// hipcc lowers __syncthreads() to a bare s_barrier when it sees no LDS/global dependency of its own, so the flag store
// could otherwise overtake the hand-over data.
__device__ __forceinline__ void chain_publish(int* flag, int value = 1) {
  if (!flag) return;      // workgroup-uniform
  stores_acked_barrier();
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// consumer side: thread 0 polls (bounded: a lost flag must not hang the GPU)
__device__ __forceinline__ void chain_wait(const int* f0, const int* f1, int* timeout) {
  if (threadIdx.x == 0) {
    int spins = 0;
    for (;;) {
      const int a0 = f0 ? __hip_atomic_load(f0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
      const int a1 = f1 ? __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1;
      if (a0 && a1) break;
      if (++spins > (1 << 17)) { if (timeout) *timeout = 1; break; }   // ~0.1 s
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();        // the consumer's loads of the handed-over data are ld_agent: no cache invalidate needed
}

// ---- merged policy-backward + policy weight-gradient launch (k_chain_bwd_pi with merge_dw): the policy chain's slices
// hand their dZ packs to the policy's dW/Adam tiles of the SAME launch. Producers store the packs write-through (sc1: the
// consumers run on other XCDs, whose L2 the data never enters before they ask for it), wait for the acknowledgements and
// bump an arrival counter; the counter has one replica per XCD (64 ints apart) so that the ~240 waiting tile workgroups
// do not all poll one word (polling one word from 480 workgroups slowed its producers down measurably in round 2).
constexpr int kArriveStride = 64;                 // ints between the replicas
__device__ __forceinline__ void pack_store4(float* p, const f32x4& v, int agent) {
  if (agent) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");   // ONE 16-byte write-through
  else nt_store4(p, v);
}
__device__ __forceinline__ void hand_store(float* p, float v, int agent) {
  if (agent) st_agent(p, v); else *p = v;
}
__device__ __forceinline__ void chain_arrive(int* cnt) {
  if (!cnt) return;        // workgroup-uniform
  stores_acked_barrier();
  if (threadIdx.x < 8) __hip_atomic_fetch_add(cnt + threadIdx.x * kArriveStride, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct ArriveWait {
  const int* cnt; int need; int* timeout;
  long long* tl = nullptr;                       // instrumented builds: chip-wide stamp when the wait ended (slot 13)
  int quick = 0;                                 // 1: short naps between polls (k_chain_bwd_qt: the waiters sit on the critical path)
  __device__ __forceinline__ void operator()() const {
    if (threadIdx.x == 0) {
      const int* c = cnt + ((int)blockIdx.x & 7) * kArriveStride;
      int spins = 0;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        if (++spins > (1 << 17)) { if (timeout) *timeout = 1; break; }   // ~0.1 s, then the hand-off word
        if (quick) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(DSACT_TILE_NAP);
      }
    }
    asm volatile("s_barrier" ::: "memory");   // the waves keep their operand loads in flight (no vmcnt wait here)
    CTLR(tl, 13);
  }
};

// block -> (unit, slice) of a forward launch; false: padding block
__device__ __forceinline__ bool fwd_decode(const FwdArgs& a, int b, int& unit, int& slice) {
  const int x = b & 7;
  unit = a.map.unit[x];
  if (unit < 0) return false;
  slice = a.map.base[x] + a.map.stride[x] * (b >> 3);
  return slice < a.u[unit].n_slices;
}
inline int fwd_grid(const FwdArgs& a) {
  int rounds = 0;
  for (int x = 0; x < 8; ++x) {
    if (a.map.unit[x] < 0) continue;
    const int n = a.u[a.map.unit[x]].n_slices - a.map.base[x];
    const int r = n > 0 ? (n + a.map.stride[x] - 1) / a.map.stride[x] : 0;
    rounds = r > rounds ? r : rounds;
  }
  return 8 * rounds;
}

// GA ("generic activation"): false compiles the GELU-only epilogue every shipped example uses -- the other activations'
// expm1f / tanhf / expf bodies cost the hot kernel registers and ~1 us per launch even when not taken
// `a`: the launch's common fields (its unit table is not read here); `u`: the unit -- an element of a.u (k_chain_fwd /
// k_chain_fwd2, kernel arguments) or of the pipelined graph's unit table in device memory (k_chain_fwdp)
// (AT / UT: FwdArgs / FwdUnit, or their constant-address-space qualified forms -- every field stays a scalar load)
// warm_cnt > 0 (pipelined launches): this workgroup is number warm_idx of the warm_cnt workgroups that run this unit on
// this XCD, and touches its share of the unit's packed weights (one dword per 128-byte line) right after its own stream
// has started: every launch begins with cold L2s (the weights were rewritten by the previous update's Adam tiles on other
// XCDs), the workgroups of a unit stream in lockstep, and 16 steps of look-ahead (~0.6 us) do not cover a miss to the
// memory side -- the first layer ran at 112 cycles per step against 81 for the later ones (profiles/r04_pipe_timeline.txt).
template <int NW, int RG, bool GA = false, typename AT = FwdArgs, typename UT = FwdUnit, bool WARM = false>
__device__ __forceinline__ void chain_fwd_body(const AT& a, const UT& u, int unit, int slice, float* lds, int warm_idx = 0, int warm_cnt = 0) {
  constexpr int W = 64 * NW, SH = W / 4, R = 4 * RG, NTHR = 64 * NW, TPR = NTHR / R;
  int* const done_flag = (u.done && !(a.debug_withhold && unit == 0 && slice == 0)) ? u.done + slice : nullptr;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane4 = lane * 4;
  const int n = 64 * wave + lane;                 // this lane's output feature
  const int row0 = slice * R;
  const int L = a.L, F = a.F, A = a.A;
  const int S0 = a.s_obs + u.s_act;               // steps of this unit's first layer
  const int head = u.head & HEAD_KIND, twin = u.head & (HEAD_TWIN_FIRST | HEAD_TWIN_SECOND);
  const int c_out = ((u.head >> 8) & 0xff) ? ((u.head >> 8) & 0xff) : W / 16;
  const int oact = GA ? head_out_act(u.head) : 0;
  const ChainLds S = chain_lds(4 * (a.s_obs + a.s_act), W, R);
  const int xin = S.off_in, red = S.off_red;
  CTL(a.timeline, 0);
  CTLR(a.timeline, 14);
  // ---- the weight stream starts before anything else
  const bool do_obs = u.seg != SEG_ACT_FROM_SAVED;
  const bool do_act = u.seg != SEG_OBS_ONLY && u.s_act > 0;
  WStr ws;
  const float* w0 = u.wf[0] + (size_t)wave * (S0 + a.tpad) * 256;
  stream_prologue(ws, w0, do_obs ? 0 : a.s_obs, lane4);
  constexpr int NWARM = 4;
  float wt[NWARM] = {0.f, 0.f, 0.f, 0.f};
  if (WARM) {   // (no branch on warm_cnt: with 0 every lane re-reads line 0 -- a branch here would cost the stream a vmcnt(0))
    const int n0 = NW * S0 * 8, nh = NW * SH * 8;                     // 128-byte lines of the first layer / of a hidden layer
    const int nhead = head == HEAD_NONE ? 0 : (head == HEAD_POLICY ? (2 * A + 15) >> 4 : 1) * (W / 16) * 8;
    const int total = warm_cnt > 0 ? n0 + (L - 1) * nh + nhead : 0;
    const int share = (total + warm_cnt - 1) / (warm_cnt > 0 ? warm_cnt : 1);
    const int lo = warm_idx * share, hi = lo + share < total ? lo + share : total;
    // (branch-free: a scalar branch between the stream prologue's loads and their use makes the compiler wait vmcnt(0))
    const float* wb[kChMaxL + 1];
#pragma unroll
    for (int l = 0; l <= kChMaxL; ++l) wb[l] = u.wf[l <= L ? l : L];
#pragma unroll
    for (int q = 0; q < NWARM; ++q) {
      int i = lo + tid + q * NTHR;
      i = i < hi ? i : lo;
      const float* base = wb[0];
      int off = i;
#pragma unroll
      for (int l = 1; l <= kChMaxL; ++l) {
        const int start = n0 + (l - 1) * nh;              // first line of layer l (the head follows the last hidden layer)
        const bool in = l <= L && i >= start;
        base = in ? wb[l] : base;
        off = in ? i - start : off;
      }
      wt[q] = *(const __attribute__((address_space(1))) float*)(base + (size_t)off * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float bq[kChMaxL];
#pragma unroll
  for (int l = 0; l < kChMaxL; ++l) bq[l] = u.bias[l < L ? l : 0][n];
  // operands of the head's row phase (TPR lanes per row, action dim d = jr, jr + TPR): fetched now -- a global load
  // issued where it is used would sit on the critical path of the kernel's tail (~1 us each round trip)
  const int mr = tid / TPR, jr = tid % TPR;
  constexpr int NQ = (32 + TPR - 1) / TPR;      // act_dim <= 32
  float pre_eps[NQ], pre_bmu[NQ], pre_braw[NQ], pre_s[NQ], pre_c[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) { pre_eps[q] = 0.f; pre_bmu[q] = 0.f; pre_braw[q] = 0.f; pre_s[q] = 1.f; pre_c[q] = 0.f; }
  if (head == HEAD_POLICY) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int d = jr + q * TPR;
      if (d < A) {
        pre_eps[q] = u.eps[(size_t)(row0 + mr) * A + d];
        pre_bmu[q] = u.bias[L][d]; pre_braw[q] = u.bias[L][A + d];
        pre_s[q] = a.act_scale[d]; pre_c[q] = a.act_center[d];
      }
    }
  } else if (head == HEAD_Q && jr == 0) {
    pre_bmu[0] = u.bias[L][0]; pre_braw[0] = u.bias[L][1];
  }
  // merged launch: the weight stream and the loads above are already in flight while this waits for its producers
  // (late_wait: only the action columns come from a producer -- the wait follows the observation segment)
  const bool pairs_in = (u.late_wait & HW_PAIRS_IN) != 0, pairs_out = (u.late_wait & HW_PAIRS_OUT) != 0;
  const unsigned tag = (pairs_in || pairs_out) ? (unsigned)(*a.tagp) + 1u : 0u;
  const bool head_wait = (u.late_wait & HW_HEAD) != 0;   // twin trunks as separate workgroups: wait0 = the partner trunk's flags,
                                                         // waited for just before the row phase
  const int* const flag0 = (pairs_in || head_wait) ? nullptr : u.wait0;      // (tagged hand-over: wait0 is the pair buffer, not a flag array)
  const bool waits = flag0 || u.wait1 || pairs_in;
  const bool late = waits && (u.late_wait & HW_LATE) && u.seg != SEG_ACT_FROM_SAVED && u.s_act > 0;
  if ((flag0 || u.wait1) && !late)
    chain_wait(flag0 ? flag0 + row0 / u.wait_rows0 : nullptr, u.wait1 ? u.wait1 + row0 / u.wait_rows1 : nullptr, a.spin_timeout);
  f32x4 zi[RG];
  if (u.seg == SEG_ACT_FROM_SAVED) {
#pragma unroll
    for (int g = 0; g < RG; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) zi[g][r] = ld_agent(u.zinit + (size_t)(row0 + 4 * g + r) * W + n);
  }
  // ---- stage the input rows: xin[r][k'] (observation part, zero padding to 4*s_obs, action part, zero padding).
  //      Four independent loads per thread and trip, stores after (a load -> wait -> store loop is one round trip per trip)
  auto stage_act = [&]() {
    const int Fp = 4 * a.s_obs;
    if (pairs_in) {
      // every lane polls ITS element until the pair carries this update's tag (bounded: a lost producer must not hang the GPU)
      const unsigned long long* hp = (const unsigned long long*)u.wait0;
      const int per_row = 4 * u.s_act;
      for (int e = tid; e < R * per_row; e += NTHR) {
        const int r = e / per_row, k = e % per_row;
        float v = 0.0f;
        if (k < A) {
          const unsigned long long* src = hp + (size_t)(row0 + r) * 32 + k;
          unsigned long long pv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int spins = 0;
          while ((unsigned)(pv >> 32) != tag) {
            if (++spins > (1 << 17)) { if (a.spin_timeout) *a.spin_timeout = 1; break; }   // ~0.1 s, then the hand-off word
            __builtin_amdgcn_s_sleep(8);
            pv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          v = __builtin_bit_cast(float, (unsigned)pv);
        }
        lds[xin + r * S.ld_in + Fp + k] = v;
      }
      return;
    }
    const int aq = u.s_act, total = R * aq;   // float4 groups of the action segment
    for (int e = tid; e < total; e += NTHR) {
      const int r = e / aq, k = (e % aq) * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const float* s = u.x + (size_t)(row0 + r) * a.ldx + F + k;
      for (int c = 0; c < 4; ++c) if (k + c < A) v[c] = ld_agent(s + c);
      *(f32x4*)(lds + xin + r * S.ld_in + Fp + k) = v;
    }
  };
  {
    const int Fp = 4 * a.s_obs;
    if (do_obs) {
      const int fq = Fp >> 2, total = R * fq;
      for (int e0 = tid; e0 < total; e0 += 4 * NTHR) {
        f32x4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = e0 + q * NTHR;
          const int r = e < total ? e / fq : 0, k = e < total ? (e % fq) * 4 : 0;
          const float* s = u.x + (size_t)(row0 + r) * a.ldx + (k + 3 < F ? k : 0);
          v[q] = *(const f32x4u*)s;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int e = e0 + q * NTHR;
          if (e < total) {
            const int r = e / fq, k = (e % fq) * 4;
            f32x4 o = v[q];
            if (k + 3 >= F) {   // row tail / padding
              const float* s = u.x + (size_t)(row0 + r) * a.ldx;
              for (int c = 0; c < 4; ++c) o[c] = k + c < F ? s[k + c] : 0.0f;
            }
            *(f32x4*)(lds + xin + r * S.ld_in + k) = o;
          }
        }
      }
    }
    if (do_act && !late) stage_act();
  }
  f32x4 acc[RG][2];
#pragma unroll
  for (int g = 0; g < RG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; }
  if (u.seg == SEG_ACT_FROM_SAVED) {
#pragma unroll
    for (int g = 0; g < RG; ++g) acc[g][0] = zi[g];
  }
  lds_barrier();
  CTL(a.timeline, 1);
  if (u.x0t) {   // the dW tiles of every first layer read the minibatch as [input feature][batch]
    const int K0 = F + (do_act ? A : 0);
    for (int e = tid; e < K0 * RG; e += NTHR) {
      const int k = e / RG, g4 = e % RG;
      const int kk = k < F ? k : 4 * a.s_obs + (k - F);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = lds[xin + (4 * g4 + r) * S.ld_in + kk];
      nt_store4(u.x0t + pk_index(k, row0 + 4 * g4, a.Cb), v);
    }
  }
  const int xs_in = xin + (lane & 3) * S.ld_in;
  // ---- first layer
  const bool more = L > 1;
  const float* w1 = u.wf[more ? 1 : 0] + (size_t)wave * (more ? SH + a.tpad : S0 + a.tpad) * 256;
  if (do_obs) {
    const bool into_act = do_act;   // continues into the action segment (same tensor, next steps) or the second layer
    gemm44_seg<RG>(ws, w0, 0, a.s_obs, into_act ? w0 : w1, into_act ? a.s_obs : 0, into_act || (more && u.seg != SEG_OBS_ONLY),
                   lds, xs_in, S.ld_in, lane4, acc);
    if (u.zsave) {
#pragma unroll
      for (int g = 0; g < RG; ++g) {
        const f32x4 v = acc[g][0] + acc[g][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) st_agent(u.zsave + (size_t)(row0 + 4 * g + r) * W + n, v[r]);
      }
      if (u.zdone) chain_publish(u.zdone + slice);   // drains this wave's prefetch queue once (~1 us) -- off the critical path
    }
    CTL(a.timeline, 2);
    if (WARM) {   // the touched values are dead; naming them here keeps the loads where they were issued
      const float wsum = (wt[0] + wt[1]) + (wt[2] + wt[3]);
      asm volatile("" : : "v"(wsum));
    }
    if (u.seg == SEG_OBS_ONLY) { CTLR(a.timeline, 15); chain_publish(done_flag); return; }
    if (u.seg == SEG_FULL_SPLIT) {   // what zsave -> zinit carries from an obs-only unit to its SEG_ACT_FROM_SAVED consumer
#pragma unroll
      for (int g = 0; g < RG; ++g) { acc[g][0] = acc[g][0] + acc[g][1]; acc[g][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
  }
  if (late) {
    if (flag0 || u.wait1)
      chain_wait(flag0 ? flag0 + row0 / u.wait_rows0 : nullptr, u.wait1 ? u.wait1 + row0 / u.wait_rows1 : nullptr, a.spin_timeout);
    stage_act();
    lds_barrier();
  }
  if (do_act) gemm44_seg<RG>(ws, w0, a.s_obs, S0, w1, 0, more, lds, xs_in, S.ld_in, lane4, acc);
  CTL(a.timeline, 3);
  // ---- epilogues + hidden layers
  NarrowFrags<4> hf;
  const int nto = head == HEAD_POLICY ? (2 * A + 15) >> 4 : 1;
  for (int l = 0; l < L; ++l) {
    if (l == L - 1 && head != HEAD_NONE) narrow_load<4>(hf, u.wf[L], c_out, nto, wave, lane4);   // under the last epilogue
    const int hn = (l & 1) ? S.off_h1 : S.off_h0;
    const float bl = l == 0 ? bq[0] : l == 1 ? bq[1] : l == 2 ? bq[2] : bq[3];
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const f32x4 z = acc[g][0] + acc[g][1] + bl;
      f32x4 hv, gd;
      if (GA) act4(u.act, z, hv, gd); else gelu4(z, hv, gd);
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[hn + (4 * g + r) * S.ld_h + n] = hv[r];
      if (u.H[l]) nt_store4(u.H[l] + pk_index(n, row0 + 4 * g, a.Cb), hv);   // rows row0+4g .. +3 of feature n: 16 contiguous bytes
      if (u.G[l]) nt_store4(u.G[l] + pk_index(n, row0 + 4 * g, a.Cb), gd);
      acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0];
    }
    lds_barrier();
    CTL(a.timeline, 4 + 2 * l);
    if (l + 1 < L) {
      const float* wc = u.wf[l + 1] + (size_t)wave * (SH + a.tpad) * 256;
      const bool has_nxt = l + 2 < L;
      const float* wn = u.wf[has_nxt ? l + 2 : l + 1] + (size_t)wave * (SH + a.tpad) * 256;
      gemm44_seg<RG>(ws, wc, 0, SH, wn, 0, has_nxt, lds, hn + (lane & 3) * S.ld_h, S.ld_h, lane4, acc);
      CTL(a.timeline, 5 + 2 * l);
    }
  }
  if (head == HEAD_NONE) { CTLR(a.timeline, 15); chain_publish(done_flag); return; }
  // ---- output layer: 16x16x4 tiles, contraction split over the waves, partials through LDS
  const int hl = ((L - 1) & 1) ? S.off_h1 : S.off_h0;
  narrow_mma<4>(hf, nto, wave, lds, hl + ((lane & 15) & (R - 1)) * S.ld_h + 4 * (lane >> 4), red, lane);
  lds_barrier();
  CTL(a.timeline, 12);
  float* const carry = lds + S.off_carry;
  if (twin == HEAD_TWIN_FIRST) {   // first trunk of a twin net: the partial outputs wait for the second trunk's
    for (int e = tid; e < R * 16 * nto; e += NTHR) {
      const int mm = e / (16 * nto), o = e % (16 * nto);
      const float v = narrow_get<4, NW>(lds, red, mm, o);
      if (u.qout) st_agent(u.qout + (size_t)(row0 + mm) * 64 + o, v);   // the partner is another workgroup
      else carry[mm * 64 + o] = v;                                       // the partner is this workgroup's next pass
    }
    CTLR(a.timeline, 15);
    if (u.qout) chain_publish(done_flag);
    return;
  }
  const bool add_carry = twin == HEAD_TWIN_SECOND;
  if (head_wait) {
    chain_wait(u.wait0 + row0 / u.wait_rows0, nullptr, a.spin_timeout);
    for (int e = tid; e < R * 16 * nto; e += NTHR) {
      const int mm = e / (16 * nto), o = e % (16 * nto);
      carry[mm * 64 + o] = ld_agent(u.zinit + (size_t)(row0 + mm) * 64 + o);
    }
    lds_barrier();
  }
  const int m = mr, j = jr;                    // row phase: TPR consecutive lanes per batch row
  const int r = row0 + m;
  if (head == HEAD_Q) {
    if (j == 0) {
      float mean = narrow_get<4, NW>(lds, red, m, 0), raw = narrow_get<4, NW>(lds, red, m, 1);
      if (add_carry) { mean += carry[m * 64]; raw += carry[m * 64 + 1]; }
      mean += pre_bmu[0]; raw += pre_braw[0];
      float sgm = 1.0f;
      if (GA && oact) { mean = out_act_fwd(oact, mean); raw = out_act_fwd(oact, raw); sgm = out_act_grad_y(oact, raw); }
      u.qout[2 * r] = mean; u.qout[2 * r + 1] = raw;
      if (u.qstd) { u.qstd[2 * r] = softplus(raw); u.qstd[2 * r + 1] = softplus_grad(raw) * sgm; }   // d std / d (pre-activation output)
    }
    CTL(a.timeline, 13);
    CTLR(a.timeline, 15);
    chain_publish(done_flag);
    return;
  }
  // policy: (mu, raw log-std) -> tanh-Gaussian rsample (act_distribution_cls.py:44-54)
  float lp = 0.f, s_tanh = 0.f, s_sig = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int d = j + q * TPR;
    if (d >= A) break;
    float mu = narrow_get<4, NW>(lds, red, m, d), raw = narrow_get<4, NW>(lds, red, m, A + d);
    if (add_carry) { mu += carry[m * 64 + d]; raw += carry[m * 64 + A + d]; }
    mu += pre_bmu[q]; raw += pre_braw[q];
    if (GA && oact) { mu = out_act_fwd(oact, mu); if (!(u.head & HEAD_STD_PLAIN)) raw = out_act_fwd(oact, raw); }
    const TanhGaussFwd f = tanh_gauss_fwd(mu, raw, pre_eps[q], pre_s[q], pre_c[q], a.lo_ls, a.hi_ls);
    lp += f.lp;
    st_agent(u.xact + (size_t)r * a.ldx + F + d, f.a);
    if (pairs_out) {
      if (!(a.debug_withhold && unit == 0 && slice == 0))
        __hip_atomic_store((unsigned long long*)u.xact2 + (size_t)r * 32 + d,
                           ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, f.a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (u.xact2) st_agent(u.xact2 + (size_t)r * a.ldx + F + d, f.a);
    u.logits[(size_t)r * 2 * A + d] = mu;
    u.logits[(size_t)r * 2 * A + A + d] = raw;
    if (!a.v1_stats) { s_tanh += tanhf(mu); s_sig += f.sigma; }
    else {
      if (d == 0) s_tanh += tanhf(mu);
      if (A >= 2) { if (d == 1) s_sig += mu; } else s_sig += f.sigma;
    }
  }
  lp = rowN_sum<TPR>(lp);
  if (j == 0) u.logp[r] = lp;
  CTL(a.timeline, 13);
  if (u.part_heads) {
    // one partial per FOUR rows: row sums over the TPR lanes of a row, then the group's rows in order -- the B/4 partial
    // sums (and the statistic k_stats forms from them) do not depend on the rows per workgroup of whoever ran this unit
    s_tanh = rowN_sum<TPR>(s_tanh); s_sig = rowN_sum<TPR>(s_sig);
    float* sc = lds + S.off_sc;     // 64 floats >= 2 * R
    if (j == 0) { sc[2 * m] = s_tanh; sc[2 * m + 1] = s_sig; }
    lds_barrier();
    if (tid < RG) {
      float t0 = 0.f, t1 = 0.f;
      for (int rr = 0; rr < 4; ++rr) { t0 += sc[2 * (4 * tid + rr)]; t1 += sc[2 * (4 * tid + rr) + 1]; }
      u.part_heads[2 * (RG * slice + tid)] = t0;
      u.part_heads[2 * (RG * slice + tid) + 1] = t1;
    }
  }
  CTLR(a.timeline, 15);
  chain_publish(done_flag);
}

template <int NW, int RG, bool GA = false>
__global__ void __launch_bounds__(64 * NW, RG >= 4 ? 1 : 2) k_chain_fwd(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int unit, slice;
  if (!fwd_decode(a, (int)blockIdx.x, unit, slice)) return;
  chain_fwd_body<NW, RG, GA>(a, a.u[unit], unit, slice, lds);
}

// Launches A and B in one: blocks [0, n_a) run group A with RGA row groups, blocks [n_a, ..) group B with RGB. Every
// group-A block has a lower id than any group-B block, so the dispatcher places all producers before any consumer, and
// producers never wait: the bounded spins of the consumers cannot deadlock. A consumer's weight stream is in flight
// while it waits; it starts the moment ITS slice's producers are done instead of after the whole of launch A plus a
// kernel boundary.
struct Fwd2Args { FwdArgs A, B; int n_a; };
static_assert(sizeof(Fwd2Args) <= 4096, "kernel arguments are limited to 4 KB");
// every unit runs 4-row (rg 1) or 8-row (rg 2) workgroups; the choice is per unit (FwdUnit::rg)
template <int NW, bool GA = false>
__global__ void __launch_bounds__(64 * NW, 2) k_chain_fwd2(Fwd2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bool in_a = (int)blockIdx.x < a.n_a;
  const FwdArgs& f = in_a ? a.A : a.B;
  int unit, slice;
  if (!fwd_decode(f, in_a ? (int)blockIdx.x : (int)blockIdx.x - a.n_a, unit, slice)) return;
  if (f.u[unit].rg == 1) chain_fwd_body<NW, 1, GA>(f, f.u[unit], unit, slice, lds);
  else chain_fwd_body<NW, 2, GA>(f, f.u[unit], unit, slice, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_fwdp: the forward launch of the PIPELINED graph (delayed-update-aware software pipelining).
// policy, log_alpha and the three target nets change only at the end of an update with it % delay_update == 0
// (dsac_v2.py:320-347), so update it + 1 of such a window sees the policy / targets update `it` saw: the launch of
// update `it` also runs pi(obs') + rsample and pi_target(obs2') + act2 / logp2 for the NEXT minibatch (resident two
// updates ahead: the riding gather looks two updates ahead in this graph), and the launch of update it + 1 then holds
// only the chains that need the fresh critics -- q1/q2(obs,act), q1/q2(obs,new_act), q1_t/q2_t(obs2,act2) -- with no
// in-launch pi -> q dependency. Units and the block -> (unit, slice) table live in DEVICE memory (one PipeFwd per
// captured launch): up to kPipeUnits units of either minibatch, each with its own rows per workgroup, XCDs and
// position in the dispatch order. A unit waits only for units that come EARLIER in every XCD's queue (the table is
// built group by group), so the bounded spins cannot deadlock. Same body, same arithmetic per row as k_chain_fwd2.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPipeUnits = 16;     // (12 roles of the pipelined graph; 16 trunk units of a merged twin-trunk forward)
constexpr int kPipeMaxBlocks = 1536;
struct PipeFwd {
  FwdArgs c;                        // common fields (c.u / c.map unused)
  FwdUnit u[kPipeUnits];
  int n_blocks;
  // role kPipeRoleBook (one block, one thread): this update's bookkeeping (prologue_duties: Adam step sizes, do_delayed) and
  // the reset of the merged critic-backward launch's arrival counters -- ahead of k_chain_bwd_qt, whose tiles read the step
  // state before they wait
  DevState* book_st; StepHyper book_hp; int* book_cnt; int book_ncnt;
  int blk[kPipeMaxBlocks];          // (unit << 16) | slice, or -1: padding block
  int warm[kPipeMaxBlocks];         // (index << 16) | count among the workgroups of the same unit on the same XCD; 0: no warm-up
};
constexpr int kPipeRoleBook = 13;
// the bookkeeping block of a forward launch (see PipeFwd::book_*): one thread
__device__ __forceinline__ void pipe_book(const __attribute__((address_space(4))) PipeFwd* p) {
  if (threadIdx.x != 0) return;
  DevState* st = (DevState*)p->book_st;
  StepHyper hp;
  hp.delay_update = p->book_hp.delay_update; hp.lr_q = p->book_hp.lr_q; hp.lr_pi = p->book_hp.lr_pi; hp.lr_alpha = p->book_hp.lr_alpha;
  hp.beta1 = p->book_hp.beta1; hp.beta2 = p->book_hp.beta2;
  prologue_duties(st, st->it_next, 1, hp);
  int* c = (int*)p->book_cnt;
  for (int i = 0; i < p->book_ncnt; ++i) c[i * kArriveStride] = 0;
}
template <int NW, bool GA = false>
__global__ void __launch_bounds__(64 * NW, 2) k_chain_fwdp(const PipeFwd* __restrict__ pd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // read-only for the whole launch: a constant-address-space view keeps every field a scalar load that no asm / atomic
  // "memory" clobber forces to be repeated (the table is written by a hipMemcpy before the graph's first launch)
  typedef __attribute__((address_space(4))) const PipeFwd KP;
  KP* p = (KP*)(unsigned long long)pd;
  const int code = p->blk[blockIdx.x];
  if (code < 0) return;
  const int unit = code >> 16, slice = code & 0xffff;
  if (unit == kPipeRoleBook) { pipe_book(p); return; }
#ifdef DSACT_TIMELINE
  if (p->c.timeline && threadIdx.x == 0 && blockIdx.x < 512) p->c.timeline[blockIdx.x * 16 + 11] = unit + 1;   // who ran here
#endif
  const int wm = p->warm[blockIdx.x];
  typedef __attribute__((address_space(4))) const FwdArgs KA;
  typedef __attribute__((address_space(4))) const FwdUnit KU;
  if (p->u[unit].rg == 1) chain_fwd_body<NW, 1, GA, KA, KU, true>(p->c, p->u[unit], unit, slice, lds, wm >> 16, wm & 0xffff);
  else chain_fwd_body<NW, 2, GA, KA, KU, true>(p->c, p->u[unit], unit, slice, lds, wm >> 16, wm & 0xffff);
}

// k_chain_fwdt: a forward launch of TWIN-trunk nets (the CNN approximators' mean / log_std MLPs over the conv features).
// Same device-memory unit / block table as k_chain_fwdp; a block code names the FIRST trunk's unit, the workgroup runs
// that unit and then unit + 1 on the same slice (see HEAD_TWIN_*). No in-launch hand-overs: group A (policy, policy
// target, critics) and group B (q targets, q(obs, new_act)) are two launches.
template <int NW, bool GA = false>
__global__ void __launch_bounds__(64 * NW, 2) k_chain_fwdt(const PipeFwd* __restrict__ pd) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __attribute__((address_space(4))) const PipeFwd KP;
  KP* p = (KP*)(unsigned long long)pd;
  const int code = p->blk[blockIdx.x];
  if (code < 0) return;
  const int unit = code >> 16, slice = code & 0xffff;
  typedef __attribute__((address_space(4))) const FwdArgs KA;
  typedef __attribute__((address_space(4))) const FwdUnit KU;
  const int nt = ((p->u[unit].head & HEAD_TWIN_FIRST) && !p->u[unit].qout) ? 2 : 1;   // (qout: the partner is another workgroup)
#pragma nounroll
  for (int t = 0; t < nt; ++t) {
    if (t) lds_barrier();
    if (p->u[unit + t].rg == 1) chain_fwd_body<NW, 1, GA, KA, KU, false>(p->c, p->u[unit + t], unit + t, slice, lds);
    else chain_fwd_body<NW, 2, GA, KA, KU, false>(p->c, p->u[unit + t], unit + t, slice, lds);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_bwd_q: loss + dZ chains of q1(obs,act), q2(obs,act), q1(obs,new_act), q2(obs,new_act)
// ---------------------------------------------------------------------------------------------------------------
// Per-thread partial sums of the critics' std column over the batch (rows tid, tid + NTHR, ...), in THAT order -- with the
// requests of four trips out before the first add (round 6): written as `for (r...) { s1 += p1[2r]; s2 += p2[2r]; }` every trip was
// load -> s_waitcnt vmcnt(0) -> add, one L2 round trip per trip at the start of every backward unit (batch 1024: 4, batch
// 4096: 16). Rows past the batch add +0.0f to a sum of positive terms: the same bits.
template <int NTHR>
__device__ __forceinline__ void std_column_sums(const float* p1, const float* p2, int B, int tid, float& s1, float& s2) {
  if (B <= NTHR) {   // one trip (the batch-256 headline): nothing to batch
    if (tid < B) { s1 += p1[2 * tid]; s2 += p2[2 * tid]; }
    return;
  }
  for (int r0 = tid; r0 < B; r0 += 4 * NTHR) {
    float v1[4], v2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = r0 + q * NTHR;
      const int rc = r < B ? r : tid;
      v1[q] = p1[2 * rc]; v2[q] = p2[2 * rc];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = r0 + q * NTHR < B;
      s1 += ok ? v1[q] : 0.0f; s2 += ok ? v2[q] : 0.0f;
    }
  }
}

struct BwdQUnit {
  const float* wb[kChMaxL];        // style-44 packed W_l^T, l = 1..L-1
  const float* wout;               // row-major output layer [2][W] of this chain's net
  const float* G[kChMaxL];         // gelu' of this chain, transposed pack [feature][batch]
  float* dZ[kChMaxL];              // transposed packs [feature][batch]
  float* dout;                     // [B][2]
  float* doutT;                    // transposed pack of dout [2 (16-row tile)][batch] (critic chains: operand of the output layer's dW)
  const float* w1at; float* dA;    // actor chains: style-16 packed (W0[:, F:])^T [16*nta x W], dL/d new_act partial [B][32]
  int which;                       // 0 q1c, 1 q2c, 2 q1p, 3 q2p
  int trunk;                       // twin-trunk nets: 0 = mean trunk (writes the shared row-phase results), 1 = log_std trunk
  float* dz0row;                   // row-major copy of dZ[0] [B][ldz0] at this trunk's columns (CNN nets: operand of dL/d features), or nullptr
  unsigned long long* dA_pairs;    // actor chains of k_chain_bwd_qpt: dL/d new_act ALSO as (value, tag) pairs [B][32] for the policy
                                   // backward slices of the same launch (nullptr: none)
};
// (the unit array comes LAST and is sized by the kernel: the merged launches carry 4 units, so that critic backward +
//  policy backward + the weight-gradient tiles' descriptors fit the 4 KB of kernel arguments)
struct BwdQTail {
  int n_units, n_slices;
  int B, A, L, Cb;
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
  int* flags_reset; int n_flags;   // the merged forward launch's ready flags: cleared here, after it and before the next one
  // DSAC_V1 (dsac_v1.py:194-253): ONE critic -- units `which` 0 (q(obs,act)) and 2 (q(obs,new_act)), no mean_std EMA, the
  // fixed TD_bound clip, and the variance-weighted pseudo-loss (v1_bound) or the Gaussian NLL; same row layout otherwise
  int v1; float td_bound; int v1_bound;
  int ldo;                         // floats between the two rows of wout (0: W; twin trunks: 2W -- the dense block-diagonal matrix)
  int c1at;                        // chunks of 16 k per 16-row tile of w1at (0: W / 16; twin trunks: 2W / 16)
  int ldz0;                        // row stride of dz0row
  int q_out_act;                   // the critics' output activation (0: linear): qout_* / qstd_* hold POST-activation values
  // merged launch (k_chain_bwd_qt): the critics' own chains (which 0 / 1) hand their dZ packs / dL/dout to the critics'
  // weight-gradient tiles of the SAME launch: counter [which] (8 replicas, kArriveStride ints apart) counts the slices of
  // that chain whose stores (write-through) have all been acknowledged
  // nullptr: not merged
  int* arrive;
  int debug_withhold;              // tests only (dsact_debug_set "withhold_flag" 3): slice 0 of q1's chain never arrives
  const long long* tagp;           // dA_pairs: DevState::tag_seq (tag = low word + 1, as in the pipelined forward launches)
};
template <int NU>
struct BwdQArgsN : BwdQTail {
  BwdQUnit u[NU];                  // twin-trunk nets: (chain, trunk) pairs -- wout / wb / G / dZ / w1at / dA point at the trunk's part
};
typedef BwdQArgsN<8> BwdQArgs;
constexpr int kBqtCntInts = 3 * 8 * 64;   // [q1, q2 chains | policy chain of k_chain_bwd_qpt][8 replicas x kArriveStride]

// row-major copy of a slice's dZ[0] from its LDS image [R][ld_h] (CNN nets: the dL/d features product reads it as a plain
// matrix). A loop of its own behind a uniform branch: the same stores inside the chains' unrolled epilogues kept 32
// address registers alive through the whole kernel (k_chain_bwd_q<4,4>: 176 -> 208 VGPRs, one wave per SIMD instead of two)
template <int W, int R, int NTHR>
__device__ __forceinline__ void store_dz0_rows(float* dst, int ldz0, int row0, const float* lds_rows, int ld_h) {
  const int tid = threadIdx.x;
  for (int e = tid; e < R * W; e += NTHR) {
    const int r = e / W, c = e - r * W;
    dst[(size_t)(row0 + r) * ldz0 + c] = lds_rows[r * ld_h + c];
  }
}

// MRG: compiled for the merged launch k_chain_bwd_qt (write-through stores + arrival counters on the critics' own chains);
// false: every hand-over branch folds away (k_chain_bwd_q is the code it was)
template <int NW, int RG, bool MRG = false, typename QA = BwdQArgs>
__device__ __forceinline__ void bwd_q_body(const QA& a, int block, float* lds) {
  if (a.flags_reset && threadIdx.x == 0)
    for (int i = block; i < a.n_flags; i += a.n_chain_blocks) a.flags_reset[i] = 0;
  int unit, slice;
  if (!chain_decode(block, a.n_units, a.n_slices, unit, slice)) return;
  constexpr int W = 64 * NW, SH = W / 4, R = 4 * RG, NTHR = 64 * NW, TPR = NTHR / R;
  const int tid = threadIdx.x;
  if (tid >= NTHR) return;                        // narrow nets: the launch is 256 wide for the riders
  const BwdQUnit& u = a.u[unit];
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane4 = lane * 4;
  const int n = 64 * wave + lane;
  const int row0 = slice * R;
  const int L = a.L;
  const ChainLds S = chain_lds(W, W, R);
  const int red = S.off_red;
  float* sc = lds + S.off_sc;
  CTL(a.timeline, 0);
  CTLR(a.timeline, 14);
  CTLV(a.timeline, 11, 1 + u.which);
  // ---- weight stream: layers L-1 .. 1 (nothing to stream for a one-hidden-layer net)
  WStr ws;
  if (L > 1) stream_prologue(ws, u.wb[L - 1] + (size_t)wave * SH * 256, 0, lane4);
  // row phase: TPR consecutive lanes per batch row
  const int m = tid / TPR, j = tid % TPR;
  const int r = row0 + m;
  // lane = hidden unit n for the output-layer backward: Wout[:, n], gelu'(z_last)[rows][n]
  const float wo0 = u.wout[n], wo1 = u.wout[(a.ldo ? a.ldo : W) + n];
  const int c1at = a.c1at ? a.c1at : W / 16;
  const bool lead = u.trunk == 0;  // the trunk units of a chain compute the same row phase; one of them publishes it
  // merged launch: this unit's dZ packs feed weight-gradient tiles of the same launch (the critics' own chains only)
  // merged launch: the critics' own chains store dZ / dL/dout write-through and, at their END (dZ[0] stored, every store
  // acknowledged), add 1 to their critic's arrival counter; every weight-gradient tile of that critic waits for it.
  // (Measured and rejected, profiles/r05_bqt_variants.txt: per-layer counters that let the later layers' tiles start while
  //  the chains still run -- with a drain of the weight stream per layer, or raised one product late behind
  //  `s_waitcnt vmcnt(kPD)`, or after the next product's first trip: the early tiles take MFMA / L2 time from the chains,
  //  whose end the first layer's 208 tiles wait for: launch 22.1 -> 22.4-24.6 us.)
  const bool merged = MRG && a.arrive != nullptr;
  const int agent = merged && u.which < 2 ? 1 : 0;
  int* const arr = agent && !(a.debug_withhold && u.which == 0 && slice == 0) ? a.arrive + u.which * (8 * kArriveStride) : nullptr;
  f32x4 gl[RG];
#pragma unroll
  for (int g = 0; g < RG; ++g) gl[g] = gload4(u.G[L - 1] + pk_index(n, row0 + 4 * g, a.Cb));
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
  // ---- batch sums of std1 / std2 -> mean_std EMA (dsac_v2.py:233-241); identical in every workgroup.
  // (BEHIND the row phase's requests, round 6: the loop's reduction waits for its loads with vmcnt(0) -- in front of them it
  //  made the start-up two memory round trips, the weight stream's prologue + the sums, then the ~20 row operands.)
  float s1 = 0.f, s2 = 0.f;
  if (a.std_sums == nullptr && !a.v1) {
    std_column_sums<NTHR>(a.qstd_c[0], a.qstd_c[1], a.B, tid, s1, s2);
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) { sc[wave] = s1; sc[4 + wave] = s2; }
  }
  lds_barrier();
  CTL(a.timeline, 1);
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
  if (a.q_out_act)   // (wave-uniform; sg1 / sg2 already carry the activation's derivative of the std output)
    d0 *= out_act_grad_y(a.q_out_act, u.which == 0 ? q1 : u.which == 1 ? q2 : u.which == 2 ? q1p : q2p);
  float v1_loss = 0.f;
  if (a.v1) {
    // dsac_v1.py:184-253, term for term as k_loss_v1 (dsact_kernels.h): one critic, z5 is q_target's sample noise
    const float qs1 = q1n + z5 * std1n;                                   // q_next_sample
    const float tq1 = rew + nd * a.gamma * (qs1 - alpha * lp2);
    const float tqb = q1 + clampf(tq1 - q1, -a.td_bound, a.td_bound);
    const float sd = fmaxf(std1, 0.0f);
    float dq = -(tq1 - q1) / (sd * sd + 0.1f);
    const float e = q1 - tqb;
    float dstd = -((e * e - sd * sd) / (sd * sd * sd + 0.1f));
    v1_loss = dq * q1 + dstd * std1;
    if (!a.v1_bound) {   // -Normal(q, std).log_prob(target_q)
      const float d = tq1 - q1, var = std1 * std1;
      dq = -d / var;
      dstd = 1.0f / std1 - (d * d) / (var * std1);
      v1_loss = (d * d) / (2.0f * var) + logf(std1) + kLogSqrt2Pi;
    }
    if (u.which == 0) { d0 = dq * a.inv_B; d1 = dstd * a.inv_B * sg1; }
    else { d0 = -a.inv_B; d1 = 0.0f; }
  }
  if (j == 0) {
    sc[16 + 2 * m] = d0; sc[16 + 2 * m + 1] = d1;
    if (lead) { u.dout[2 * r] = d0; u.dout[2 * r + 1] = d1; }
    if (lead && u.doutT) { hand_store(u.doutT + pk_index(0, r, a.Cb), d0, agent); hand_store(u.doutT + pk_index(1, r, a.Cb), d1, agent); }
    if (!lead) {
    } else if (u.which == 0 && a.v1) {
      float* pl = a.part_loss + (size_t)r * kLossPart;
      pl[0] = v1_loss; pl[1] = 0.f; pl[2] = q1; pl[3] = 0.f; pl[4] = std1; pl[5] = 0.f;
      pl[6] = alpha * lpn - q1p;
      pl[7] = lpn;
      pl[8] = r == 0 ? alpha : 0.0f;
      pl[9] = 0.0f; pl[10] = std1; pl[11] = std1;
      if (r == 0) { hand_store(a.grads_tail, 0.f, merged); hand_store(a.grads_tail + 1, 0.f, merged); }
    } else if (u.which == 0) {
      float* pl = a.part_loss + (size_t)r * kLossPart;
      pl[0] = c1.loss; pl[1] = c2.loss; pl[2] = q1; pl[3] = q2; pl[4] = std1; pl[5] = std2;
      pl[6] = alpha * lpn - fminf(q1p, q2p);
      pl[7] = lpn;
      pl[8] = r == 0 ? alpha : 0.0f;
      pl[9] = 0.0f; pl[10] = std1; pl[11] = std2;
      if (r == 0) { hand_store(a.grads_tail, ms1, merged); hand_store(a.grads_tail + 1, ms2, merged); }   // (the closing block of a merged launch reads them)
    }
  }
  lds_barrier();
  // ---- dZ of the last hidden layer: (dOut . Wout) * gelu'   (lane = hidden unit, registers = batch rows)
#pragma unroll
  for (int g = 0; g < RG; ++g) {
    f32x4 ov;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) ov[rr] = (sc[16 + 2 * (4 * g + rr)] * wo0 + sc[16 + 2 * (4 * g + rr) + 1] * wo1) * gl[g][rr];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) lds[S.off_h0 + (4 * g + rr) * S.ld_h + n] = ov[rr];
    pack_store4(u.dZ[L - 1] + pk_index(n, row0 + 4 * g, a.Cb), ov, agent);
  }
  NarrowFrags<2> af;
  const int nta = (a.A + 15) >> 4;
  if (u.w1at && L == 1 && !MRG) narrow_load<2>(af, u.w1at, c1at, nta, wave, lane4);
  if (arr && L == 1) chain_arrive(arr);   // one hidden layer: dZ[0] is the last thing this chain delivers
  else lds_barrier();
  if (u.dz0row && L == 1) store_dz0_rows<W, R, NTHR>(u.dz0row, a.ldz0, row0, lds + S.off_h0, S.ld_h);
  CTL(a.timeline, 2);
  // ---- hidden layers: dZ[l-1] = (dZ[l] W_l) * gelu'(z[l-1])
  f32x4 acc[RG][2];
  int cur = 0;
  for (int l = L - 1; l >= 1; --l) {
#pragma unroll
    for (int g = 0; g < RG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; }
    f32x4 gq[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) gq[g] = gload4(u.G[l - 1] + pk_index(n, row0 + 4 * g, a.Cb));
    const bool has_nxt = l > 1;
    const float* wl = u.wb[l] + (size_t)wave * SH * 256;
    const float* wn = u.wb[has_nxt ? l - 1 : l] + (size_t)wave * SH * 256;
    const int xop = (cur ? S.off_h1 : S.off_h0) + (lane & 3) * S.ld_h;
    gemm44_seg<RG>(ws, wl, 0, SH, wn, 0, has_nxt, lds, xop, S.ld_h, lane4, acc);
    CTL(a.timeline, 3 + 2 * (L - 1 - l));
    // (merged launch: the 32 fragment registers of dL/da are loaded AFTER the last epilogue -- one L2 round trip on a chain
    //  whose end nobody in this launch waits for -- so that the kernel fits 168 registers: three workgroups per CU, the
    //  chain's and two waiting tiles)
    if (l == 1 && u.w1at && !MRG) narrow_load<2>(af, u.w1at, c1at, nta, wave, lane4);
    const int hn = cur ? S.off_h0 : S.off_h1;
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const f32x4 dz = (acc[g][0] + acc[g][1]) * gq[g];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) lds[hn + (4 * g + rr) * S.ld_h + n] = dz[rr];
      pack_store4(u.dZ[l - 1] + pk_index(n, row0 + 4 * g, a.Cb), dz, agent);
    }
    cur ^= 1;
    if (arr && l == 1) chain_arrive(arr);                // dZ[0]: explicit acknowledgement wait (the chain ends here)
    else lds_barrier();
    if (u.dz0row && l == 1) store_dz0_rows<W, R, NTHR>(u.dz0row, a.ldz0, row0, lds + hn, S.ld_h);
    CTL(a.timeline, 4 + 2 * (L - 1 - l));
  }
  if (!u.w1at) { CTLR(a.timeline, 15); return; }
  if (MRG) narrow_load<2>(af, u.w1at, c1at, nta, wave, lane4);
  // ---- dL/d new_act through this critic: dZ0 . W0[:, F:F+A]   (contraction over the hidden units, split over waves)
  narrow_mma<2>(af, nta, wave, lds, (cur ? S.off_h1 : S.off_h0) + ((lane & 15) & (R - 1)) * S.ld_h + 4 * (lane >> 4), red, lane);
  lds_barrier();
  const unsigned ptag = (MRG && u.dA_pairs) ? (unsigned)(*a.tagp) + 1u : 0u;
  for (int d = j; d < 16 * nta; d += TPR) {
    const float v = d < a.A ? narrow_get<2, NW>(lds, red, m, d) : 0.0f;
    u.dA[(size_t)r * 32 + d] = v;
    if (MRG && u.dA_pairs)   // the data is the flag: one 8-byte agent-scope store per element (k_chain_bwd_qpt's policy slices poll them)
      __hip_atomic_store(u.dA_pairs + (size_t)r * 32 + d, ((unsigned long long)ptag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v),
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  CTL(a.timeline, 12);
  CTLR(a.timeline, 15);
}

template <int NW, int RG>
__global__ void __launch_bounds__(256) k_chain_bwd_q(BwdQArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_chain_blocks) { loss_rider(a.ride); return; }   // riders are 256-thread blocks
  bwd_q_body<NW, RG>(a, (int)blockIdx.x, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_bwd_pi: dL/d new_act -> rsample backward -> policy output-layer backward -> policy dZ chain
// blocks >= n_chain_blocks: ride-along weight-gradient tiles (the critics' dW + Adam), 256 threads each
// ---------------------------------------------------------------------------------------------------------------
struct BwdPiArgs {
  // twin-trunk policy (CNN nets): n_trunks = 2 -- chain blocks [0, n_chain_blocks / 2) run the mean trunk (woutT, wb, G, dZ),
  // the rest the log_std trunk (woutT1, wb1, G1, dZ1); both compute the row phase, the first publishes it. dA2: the
  // log_std trunks' share of dL/d new_act (nullptr: none)
  int n_trunks;
  const float* woutT1; const float* wb1[kChMaxL]; const float* G1[kChMaxL]; float* dZ1[kChMaxL];
  const float* dA2[2];
  float* dz0row; int ldz0;         // row-major copy of the policy's dZ[0] [B][ldz0] (CNN nets: operand of dL/d features), or nullptr
  const float* dA[2];              // [B][32] from k_chain_bwd_q (q1p, q2p)
  const float* logits_pi; const float* eps_new; const float* log_alpha;
  const float* woutT; int SoT;     // style-44 packed Wout_pi^T [W x 4*SoT]
  const float* wb[kChMaxL];        // style-44 packed policy layers (transposed)
  const float* G[kChMaxL];         // transposed packs [feature][batch]
  float* dZ[kChMaxL];
  float* dout_pi; float* d_new_act;
  float* dout_piT;                 // transposed pack of dout_pi [2A (16-row tiles)][batch]
  int n_slices, B, A, L, Cb;
  float inv_B; int auto_alpha; float alpha_fixed;
  int pi_out_act, pi_out_n;        // the policy's output activation (0: linear) and the outputs it applies to (A: mean half only); logits_pi is POST-activation
  const float* act_scale; float lo_ls, hi_ls;
  const float* part_loss; int n_part; float target_entropy; float* grad_log_alpha;
  int n_chain_blocks;
  int tile0, n_extra;              // riders: weight-gradient tiles [tile0, tile0 + n_extra) of `dw`
  // merged launch (dw_chunks == 1): the policy's own tiles [pi_tile0, pi_tile0 + n_pi_tiles) follow the riders and wait for
  // the arrival counter of the chain's slices; one more block closes the update (alpha gradient, finalize_update)
  int merge_dw, pi_tile0, n_pi_tiles, finalize;
  int* cnt_pi; int* spin_timeout;
  // k_chain_bwd_qpt: dL/d new_act arrives from the critics' chains of the SAME launch as (value, tag) pairs [B][32] per critic
  // (every lane polls ITS elements until they carry this update's tag; bounded). nullptr: plain loads of dA (an earlier launch)
  const unsigned long long* dA_pairs[2]; const long long* tagp;
  int debug_withhold;              // tests only (dsact_debug_set "withhold_flag" 2): slice 0 never arrives
  long long* timeline;
  Dw2Args dw;
};

__device__ __forceinline__ void bwd_pi_alpha_grad(const BwdPiArgs& a, int lane) {
  float s = 0.f;
  for (int r0 = 0; r0 < a.n_part; r0 += 64) {
    const int rr = r0 + lane;
    s += rr < a.n_part ? a.part_loss[(size_t)rr * kLossPart + 7] : 0.f;
  }
  s = wave_sum(s);
  if (lane == 0) a.grad_log_alpha[0] = a.auto_alpha ? -(s * a.inv_B + a.target_entropy) : 0.0f;
}

template <int NW, int RG>
__device__ __forceinline__ void bwd_pi_body(const BwdPiArgs& a, int slice, float* lds, int trunk = 0) {
  constexpr int W = 64 * NW, SH = W / 4, R = 4 * RG, NTHR = 64 * NW, TPR = NTHR / R;
  const int tid = threadIdx.x;
  if (slice >= a.n_slices || tid >= NTHR) return;
  const bool lead = trunk == 0;
  const float* const* const t_wb = trunk ? a.wb1 : a.wb;
  const float* const* const t_G = trunk ? a.G1 : a.G;
  float* const* const t_dZ = trunk ? a.dZ1 : a.dZ;
  float* const dz0row = a.dz0row ? a.dz0row + trunk * W : nullptr;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane4 = lane * 4;
  const int n = 64 * wave + lane;
  const int row0 = slice * R;
  const int L = a.L, A = a.A;
  const ChainLds S = chain_lds(4 * a.SoT, W, R);
  float* xdo = lds + S.off_in;
  CTL(a.timeline, 0);
  CTLR(a.timeline, 14);
  WStr ws;
  const float* wo = (trunk ? a.woutT1 : a.woutT) + (size_t)wave * a.SoT * 256;
  stream_prologue(ws, wo, 0, lane4);
  const int m = tid / TPR, j = tid % TPR;
  const int r = row0 + m;
  // the row phase's inputs, fetched before anything waits (each would otherwise be a round trip on the critical path)
  constexpr int NQ = (32 + TPR - 1) / TPR;      // act_dim <= 32
  float pdA[NQ], pmu[NQ], praw[NQ], peps[NQ], psc[NQ];
  const bool pairs = a.dA_pairs[0] != nullptr;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int d = j + q * TPR;
    const bool ok = d < A;
    pdA[q] = (ok && !pairs) ? a.dA[0][(size_t)r * 32 + d] + a.dA[1][(size_t)r * 32 + d] : 0.f;
    if (a.dA2[0] && ok) pdA[q] += a.dA2[0][(size_t)r * 32 + d] + a.dA2[1][(size_t)r * 32 + d];
    pmu[q] = ok ? a.logits_pi[(size_t)r * 2 * A + d] : 0.f;
    praw[q] = ok ? a.logits_pi[(size_t)r * 2 * A + A + d] : 0.f;
    peps[q] = ok ? a.eps_new[(size_t)r * A + d] : 0.f;
    psc[q] = ok ? a.act_scale[d] : 1.f;
  }
  if (pairs) {
    // the critics' chains of this launch deliver dL/d new_act as (value, tag) pairs: this slice's weight stream and every other
    // input are already in flight while its lanes poll their own elements (same sum, same order as the plain loads)
    const unsigned tag = (unsigned)(*a.tagp) + 1u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int d = j + q * TPR;
      if (d < A) {
        float v[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned long long* src = a.dA_pairs[c] + (size_t)r * 32 + d;
          unsigned long long pv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          int spins = 0;
          while ((unsigned)(pv >> 32) != tag) {
            if (++spins > (1 << 17)) { if (a.spin_timeout) *a.spin_timeout = 1; break; }   // ~0.1 s, then the hand-off word
            __builtin_amdgcn_s_sleep(8);
            pv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          v[c] = __builtin_bit_cast(float, (unsigned)pv);
        }
        pdA[q] = v[0] + v[1];
      }
    }
  }
  // alpha gradient (dsac_v2.py:312-318): -mean(logp_new + target_entropy)
  if (slice == 0 && wave == 0 && !a.merge_dw && lead) bwd_pi_alpha_grad(a, lane);   // merged launch: the closing block does it
  const int agent = a.merge_dw;
  const float alpha = a.auto_alpha ? expf(a.log_alpha[0]) : a.alpha_fixed;
  // zero the operand rows (padding included), then fill (dmu | draw)
  for (int e = tid; e < R * 4 * a.SoT; e += NTHR) xdo[(e / (4 * a.SoT)) * S.ld_in + e % (4 * a.SoT)] = 0.0f;
  lds_barrier();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int d = j + q * TPR;
    if (d >= A) break;
    const float dA = pdA[q];
    float dmu, draw;
    tanh_gauss_bwd(pmu[q], praw[q], peps[q], psc[q], a.lo_ls, a.hi_ls, dA, alpha * a.inv_B, dmu, draw);
    if (a.pi_out_act) {   // (wave-uniform)
      dmu *= out_act_grad_y(a.pi_out_act, pmu[q]);
      if (A + d < a.pi_out_n) draw *= out_act_grad_y(a.pi_out_act, praw[q]);
    }
    if (lead) {
      a.dout_pi[(size_t)r * 2 * A + d] = dmu;
      a.dout_pi[(size_t)r * 2 * A + A + d] = draw;
      hand_store(a.dout_piT + pk_index(d, r, a.Cb), dmu, agent);
      hand_store(a.dout_piT + pk_index(A + d, r, a.Cb), draw, agent);
      a.d_new_act[(size_t)r * A + d] = dA;
    }
    xdo[m * S.ld_in + d] = dmu;
    xdo[m * S.ld_in + A + d] = draw;
  }
  lds_barrier();
  CTL(a.timeline, 1);
  f32x4 acc[RG][2];
  // ---- policy output layer backward: (dmu | draw) . Wout, then * gelu'(z_last)
  {
#pragma unroll
    for (int g = 0; g < RG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; }
    f32x4 gq[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) gq[g] = gload4(t_G[L - 1] + pk_index(n, row0 + 4 * g, a.Cb));
    const bool has_nxt = L > 1;
    gemm44_seg<RG>(ws, wo, 0, a.SoT, has_nxt ? t_wb[L - 1] + (size_t)wave * SH * 256 : wo, 0, has_nxt,
                   lds, S.off_in + (lane & 3) * S.ld_in, S.ld_in, lane4, acc);
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const f32x4 dz = (acc[g][0] + acc[g][1]) * gq[g];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) lds[S.off_h0 + (4 * g + rr) * S.ld_h + n] = dz[rr];
      pack_store4(t_dZ[L - 1] + pk_index(n, row0 + 4 * g, a.Cb), dz, agent);
    }
    lds_barrier();
    if (dz0row && L == 1) store_dz0_rows<W, R, NTHR>(dz0row, a.ldz0, row0, lds + S.off_h0, S.ld_h);
    CTL(a.timeline, 2);
  }
  int cur = 0;
  for (int l = L - 1; l >= 1; --l) {
#pragma unroll
    for (int g = 0; g < RG; ++g) { acc[g][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[g][1] = acc[g][0]; }
    f32x4 gq[RG];
#pragma unroll
    for (int g = 0; g < RG; ++g) gq[g] = gload4(t_G[l - 1] + pk_index(n, row0 + 4 * g, a.Cb));
    const bool has_nxt = l > 1;
    gemm44_seg<RG>(ws, t_wb[l] + (size_t)wave * SH * 256, 0, SH, t_wb[has_nxt ? l - 1 : l] + (size_t)wave * SH * 256, 0, has_nxt,
                   lds, (cur ? S.off_h1 : S.off_h0) + (lane & 3) * S.ld_h, S.ld_h, lane4, acc);
    const int hn = cur ? S.off_h0 : S.off_h1;
#pragma unroll
    for (int g = 0; g < RG; ++g) {
      const f32x4 dz = (acc[g][0] + acc[g][1]) * gq[g];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) lds[hn + (4 * g + rr) * S.ld_h + n] = dz[rr];
      pack_store4(t_dZ[l - 1] + pk_index(n, row0 + 4 * g, a.Cb), dz, agent);
    }
    cur ^= 1;
    if (l > 1 || dz0row) lds_barrier();
    if (dz0row && l == 1) store_dz0_rows<W, R, NTHR>(dz0row, a.ldz0, row0, lds + hn, S.ld_h);
    CTL(a.timeline, 3 + (L - 1 - l));
  }
  CTLR(a.timeline, 15);
  if (a.merge_dw && !(a.debug_withhold && slice == 0)) chain_arrive(a.cnt_pi);
}

// the arrival counter the policy's tiles wait for: ONE, raised at the end of the chain (per-layer counters -- tiles of later
// layers starting while the chain still runs -- were measured slower in round 5, 22.5-22.8 -> 23.4-23.6 us: what the early
// tiles move through L2 / the fabric slows the chain whose END the first layer's 96 tiles wait for; profiles/r05_bqt_variants.txt)
__device__ __forceinline__ const int* pi_tile_counter(const BwdPiArgs& a, int) { return a.cnt_pi; }

// blocks past the chain's: riders (xcd_chunk_grid(n_extra) per batch range), then -- merged launch -- the policy's
// tiles and the closing block
__device__ __forceinline__ void bwd_pi_tail_blocks(const BwdPiArgs& a, int idx, float* lds) {
  const int per_range = xcd_chunk_grid(a.n_extra);   // n_chain_blocks is a multiple of 8: riders start on XCD 0
  const int n_rider_blocks = a.merge_dw ? per_range : 0x7fffffff;   // (unmerged: every remaining block is a rider, ranges in y)
  int t;
  if (idx < n_rider_blocks) {
    if (!xcd_chunk(idx % per_range, a.n_extra, t)) return;
    CTLR(a.timeline, 14);
    dw2_tile(a.dw, (idx / per_range) * a.dw.n_base + a.tile0 + t, lds);
    CTLR(a.timeline, 15);
    return;
  }
  idx -= n_rider_blocks;
  if (idx < xcd_chunk_grid(a.n_pi_tiles)) {
    if (!xcd_chunk(idx, a.n_pi_tiles, t)) return;
    dw2_tile<2, ArriveWait>(a.dw, a.pi_tile0 + t, lds, ArriveWait{pi_tile_counter(a, a.pi_tile0 + t), a.n_slices, a.spin_timeout});
    return;
  }
  // closing block: waits for the chain as well -- its slices read log_alpha (alpha = exp(log_alpha)) when they start, and
  // this block's Adam step on log_alpha must not overtake them (found by the merged == split test at a small shape, where
  // the block is dispatched within a microsecond of the chain). The alpha gradient reads the loss launch's partial sums (an
  // earlier launch); the step state finalize_update commits is not what the tiles read.
  ArriveWait{a.cnt_pi, a.n_slices, a.spin_timeout}();
  if (threadIdx.x < 64) {
    bwd_pi_alpha_grad(a, (int)threadIdx.x);
    if (a.finalize && threadIdx.x == 0) finalize_update(a.dw.fo);
  }
}

template <int NW, int RG>
__global__ void __launch_bounds__(256) k_chain_bwd_pi(BwdPiArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if ((int)blockIdx.x >= a.n_chain_blocks) { bwd_pi_tail_blocks(a, (int)blockIdx.x - a.n_chain_blocks, lds); return; }
  const int per = a.n_chain_blocks >> 1, b = (int)blockIdx.x;
  const int trunk = (a.n_trunks == 2 && b >= per) ? 1 : 0;   // (one inlined body: the trunk's pointers are scalar selects)
  bwd_pi_body<NW, RG>(a, trunk ? b - per : b, lds, trunk);
}
// (Round 3-5 kept a 512-thread form of this launch whose riding tiles ran eight waves, DSACT_RIDE8: measured equal at batch
//  >= 1024 -- 27.4 vs 27.1 us, 8,784 vs 8,767 steps/s: that launch is not bound by the riders' wave count -- and removed in round 6.)

// ---------------------------------------------------------------------------------------------------------------
// k_chain_bwd_qt (round 5): the critics' backward AND their weight-gradient / Adam tiles AND the block that closes the update
// in ONE launch -- the last two launches of an update that leaves the policy alone in the pipelined graph (chain_bwd_q 12.3 us
// + chain_dw_q 11.7 us: the tiles' kernel boundary, dispatch ramp and first-touch latency sat on the critics' dependent chain
// forward -> backward -> dW + Adam -> next forward). The critics' own chains (q1c, q2c) store their dZ packs and dL/dout
// write-through and raise their critic's arrival counter when they end (BwdQArgs::arrive); every tile of that critic waits
// for it with its X-side fragments and Adam operands already in flight (dw2_tile's `wait`): all 480 tiles are resident beside
// the chains from the start (144 registers: three workgroups per CU) and run the moment the chains end. Blocks: [chain
// slices] [riders: the gather of minibatch s + 2] [tiles, dealt to the XCDs layer class by layer class] [closing block]. This update's bookkeeping (Adam step sizes: the tiles read them BEFORE they wait) moved into the
// forward launch of the same update (k_chain_fwdp, role kPipeRoleBook), which also zeroes the arrival counters. Every wait
// targets blocks with lower ids (resident first), bounded like the other in-launch hand-overs. Same arithmetic per tile and
// per row as the two launches: bit-identical (tests/test_hip_parity.py::test_pipelined_graph_equals_eager_steps).
// ---------------------------------------------------------------------------------------------------------------
struct BwdQtArgs {
  BwdQArgsN<4> q;
  Dw2Args dw;
  const int* tile_tab; int n_tile_blocks;      // block behind the chain slices and riders -> base tile of dw, or -1 (padding)
  int n_riders;
  int need;                                    // slices of one critic chain
  int* spin_timeout;
  const float* logp_new; int n_part; float target_entropy; float* grad_log_alpha; int finalize;
};
static_assert(sizeof(BwdQtArgs) <= 4096, "kernel arguments are limited to 4 KB");

#ifndef DSACT_BQT_OCC
#define DSACT_BQT_OCC 3
#endif
template <int NW, int RG>
__global__ void __launch_bounds__(256, DSACT_BQT_OCC) k_chain_bwd_qt(BwdQtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = (int)blockIdx.x;
  if (b < a.q.n_chain_blocks) { bwd_q_body<NW, RG, true, BwdQArgsN<4> >(a.q, b, lds); return; }
  int idx = b - a.q.n_chain_blocks;
  if (idx < a.n_riders) { loss_rider(a.q.ride); return; }
  idx -= a.n_riders;
  const int L = a.q.L;
  if (idx < a.n_tile_blocks) {
    const int t = ((const __attribute__((address_space(4))) int*)(unsigned long long)a.tile_tab)[idx];
    if (t < 0) return;
    // problem of the tile -> (critic, layer): problems are [critic][layer 0 .. L-1, output layer]
    int pi = 0;
#pragma unroll
    for (int q = 0; q + 1 < kMaxDwProb; ++q)
      if (q + 1 < a.dw.n_prob && t >= a.dw.tile_ends[q]) pi = q + 1;
    const int net = pi / (L + 1), l = pi - net * (L + 1);
    const int* cnt = a.q.arrive + net * 8 * kArriveStride;
    CTLR(a.q.timeline, 14);
    CTLV(a.q.timeline, 11, 10 + (l < L ? l : L - 1));   // tile class
    dw2_tile<2, ArriveWait>(a.dw, t, lds, ArriveWait{cnt, a.need, a.spin_timeout, a.q.timeline, 0});
    CTLR(a.q.timeline, 15);
    return;
  }
  // closing block: the chains have passed their row phase (mean_std tail) and nothing reads the step state any more once
  // every slice of both critics has delivered its last layer
  ArriveWait{a.q.arrive, a.need, a.spin_timeout}();
  if (a.q.n_units > 1 && !a.q.v1) ArriveWait{a.q.arrive + 8 * kArriveStride, a.need, a.spin_timeout}();
  if (threadIdx.x < 64) {
    // alpha gradient (dsac_v2.py:312-318): -mean(logp_new + target_entropy). part_loss[.][7] IS logp_new (bwd_q_body's row
    // phase copies it): read from the forward launch's buffer, in bwd_pi_alpha_grad's order
    const int lane = (int)threadIdx.x;
    float s = 0.f;
    for (int r0 = 0; r0 < a.n_part; r0 += 64) {
      const int rr = r0 + lane;
      s += rr < a.n_part ? a.logp_new[rr] : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) {
      a.grad_log_alpha[0] = a.q.auto_alpha ? -(s * a.q.inv_B + a.target_entropy) : 0.0f;
      if (a.finalize) finalize_update(a.dw.fo);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_bwd_qpt (round 5): the WHOLE backward of an update that moves the policy (iteration % delay_update == 0) in the
// pipelined graph as ONE launch -- critics' backward chains -> (dL/d new_act as tagged pairs) -> policy backward chain ->
// the policy's weight-gradient / Adam / Polyak tiles; the critics' tiles wait for the critics' own chains (as in
// k_chain_bwd_qt) and run while the policy chain works; one block closes the update. Replaces chain_bwd_q (12.6 us) +
// chain_bwd_pi (22.5 us): the policy chain's slices are resident from the start with their weight stream, gelu' packs and
// row-phase inputs in flight, so the kernel boundary, the dispatch ramp and their ~3.4 us of start-up leave the dependent
// chain forward -> dL/da -> policy backward -> policy Adam -> next forward. Blocks: [critic chain slices] [policy chain
// slices] [riders: gather of minibatch s + 2] [critics' tiles (table)] [policy tiles] [closing block]; every wait targets
// blocks with lower ids. Bookkeeping and the reset of the three arrival counters ride in this update's forward launch.
// ---------------------------------------------------------------------------------------------------------------
struct BwdQpArgs {
  BwdQArgsN<4> q;
  BwdPiArgs pi;                                // pi.dw: the problem list of ALL weight-gradient tiles of the launch
  const int* tile_tab; int n_tile_blocks;      // the critics' tiles (k_chain_bwd_qt's table)
  int n_riders;
  int need_c;                                  // slices of one critic chain
  int* spin_timeout;
  const float* logp_new; int n_part; float target_entropy; float* grad_log_alpha; int finalize;
};
static_assert(sizeof(BwdQpArgs) <= 4096, "kernel arguments are limited to 4 KB");

template <int NW, int RGQ, int RGP>
__global__ void __launch_bounds__(256, DSACT_BQT_OCC) k_chain_bwd_qpt(BwdQpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int idx = (int)blockIdx.x;
  if (idx < a.q.n_chain_blocks) { bwd_q_body<NW, RGQ, true, BwdQArgsN<4> >(a.q, idx, lds); return; }
  idx -= a.q.n_chain_blocks;
  if (idx < a.pi.n_chain_blocks) { CTLV(a.pi.timeline, 11, 5); bwd_pi_body<NW, RGP>(a.pi, idx, lds); return; }
  idx -= a.pi.n_chain_blocks;
  if (idx < a.n_riders) { loss_rider(a.q.ride); return; }
  idx -= a.n_riders;
  const int L = a.q.L;
  const int n_pol_blocks = xcd_chunk_grid(a.pi.n_pi_tiles);
  if (idx < a.n_tile_blocks + n_pol_blocks) {
    // ONE call site of dw2_tile for both tile kinds (two inlined copies made hipcc keep the whole kernel-argument struct in
    // scratch: 3.7 KB per lane): the critics' tiles come through the table and wait for their critic's chain, the policy's
    // through xcd_chunk and wait for the policy chain
    int t;
    const int* cnt;
    int need;
    if (idx < a.n_tile_blocks) {
      t = ((const __attribute__((address_space(4))) int*)(unsigned long long)a.tile_tab)[idx];
      if (t < 0) return;
      int pi = 0;
#pragma unroll
      for (int q = 0; q + 1 < kMaxDwProb; ++q)
        if (q + 1 < a.pi.dw.n_prob && t >= a.pi.dw.tile_ends[q]) pi = q + 1;
      cnt = a.q.arrive + (pi / (L + 1)) * 8 * kArriveStride;
      need = a.need_c;
      CTLV(a.q.timeline, 11, 10);      // critics' tile
    } else {
      if (!xcd_chunk(idx - a.n_tile_blocks, a.pi.n_pi_tiles, t)) return;
      t += a.pi.pi_tile0;
      cnt = a.pi.cnt_pi;
      need = a.pi.n_slices;
      CTLV(a.q.timeline, 11, 11);      // policy tile
    }
    CTLR(a.q.timeline, 14);
    dw2_tile<2, ArriveWait>(a.pi.dw, t, lds, ArriveWait{cnt, need, a.spin_timeout, a.q.timeline});
    CTLR(a.q.timeline, 15);
    return;
  }
  // closing block: the policy chain's slices read log_alpha when they start (this block's Adam step on it must not overtake
  // them) and the critics' chains wrote the mean_std tail
  ArriveWait{a.pi.cnt_pi, a.pi.n_slices, a.spin_timeout}();
  ArriveWait{a.q.arrive, a.need_c, a.spin_timeout}();
  ArriveWait{a.q.arrive + 8 * kArriveStride, a.need_c, a.spin_timeout}();
  if (threadIdx.x < 64) {
    const int lane = (int)threadIdx.x;
    float s = 0.f;
    for (int r0 = 0; r0 < a.n_part; r0 += 64) {
      const int rr = r0 + lane;
      s += rr < a.n_part ? a.logp_new[rr] : 0.f;
    }
    s = wave_sum(s);
    if (lane == 0) {
      a.grad_log_alpha[0] = a.q.auto_alpha ? -(s * a.q.inv_B + a.target_entropy) : 0.0f;
      if (a.finalize) finalize_update(a.pi.dw.fo);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_chain_fwdpb: k_chain_fwdp + the DEFERRED policy backward of the previous update. An update that leaves the policy
// alone (iteration % delay_update != 0) still computes the policy gradient -- the reference does (dsac_v2.py:174-186)
// and discards it (:324) -- but nothing reads it: its rsample backward, policy dZ chain and the policy's 240 weight-gradient
// tiles were the critical path of that update's last launch (chain 12 us, then the tiles: 20 us) for no consumer. In the
// pipelined graph they ride in the NEXT update's forward launch, which holds only the fresh-critic chains and leaves a
// slot per CU free: same arithmetic, same buffers (written before that update's own policy backward needs them), and the
// previous update's last launch is the critics' weight-gradient / Adam tiles + the block that closes the update.
// Block table codes: unit kPipeRoleBwdPi = a slice of the policy backward chain, kPipeRoleTile = policy tile `slice`
// (waits for the chain's arrival counter like the merged policy-backward launch). 256 threads wide for the tiles.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPipeRoleBwdPi = 14, kPipeRoleTile = 15;
template <int NW, bool GA = false>
__global__ void __launch_bounds__(256, 2) k_chain_fwdpb(const PipeFwd* __restrict__ pd, BwdPiArgs bp, int bp_rg) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __attribute__((address_space(4))) const PipeFwd KP;
  KP* p = (KP*)(unsigned long long)pd;
  const int code = p->blk[blockIdx.x];
  if (code < 0) return;
  const int unit = code >> 16, slice = code & 0xffff;
#ifdef DSACT_TIMELINE
  if (p->c.timeline && threadIdx.x == 0 && blockIdx.x < 512) p->c.timeline[blockIdx.x * 16 + 11] = unit + 1;
#endif
  if (unit == kPipeRoleBwdPi) {
    if (bp_rg == 1) bwd_pi_body<NW, 1>(bp, slice, lds); else bwd_pi_body<NW, 2>(bp, slice, lds);
    return;
  }
  if (unit == kPipeRoleTile) {
    dw2_tile<2, ArriveWait>(bp.dw, bp.pi_tile0 + slice, lds, ArriveWait{pi_tile_counter(bp, bp.pi_tile0 + slice), bp.n_slices, bp.spin_timeout});
    return;
  }
  if (unit == kPipeRoleBook) { pipe_book(p); return; }
  if ((int)threadIdx.x >= 64 * NW) return;      // narrow nets: the launch is 256 wide for the tiles
  const int wm = p->warm[blockIdx.x];
  typedef __attribute__((address_space(4))) const FwdArgs KA;
  typedef __attribute__((address_space(4))) const FwdUnit KU;
  if (p->u[unit].rg == 1) chain_fwd_body<NW, 1, GA, KA, KU, true>(p->c, p->u[unit], unit, slice, lds, wm >> 16, wm & 0xffff);
  else chain_fwd_body<NW, 2, GA, KA, KU, true>(p->c, p->u[unit], unit, slice, lds, wm >> 16, wm & 0xffff);
}

}  // namespace dsact
