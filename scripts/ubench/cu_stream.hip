// How fast can ONE compute unit pull a read-once stream (weights) out of L2 / Infinity Cache?
// Each workgroup reads `kb` KB in 1 KB-per-wave-instruction pieces (fragment-major "packed" layout) and sums them.
// Variants: plain global_load_dwordx4, nontemporal, sc1 (L1 bypass), LDS-DMA (global_load_lds_dwordx4) + ds_read.
// usage: cu_stream [n_blocks=64]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const f32x4 gf4;

template <int MODE>
__device__ __forceinline__ f32x4 ld(const float* p) {
  if (MODE == 0) return *(gf4*)p;
  if (MODE == 1) return __builtin_nontemporal_load((gf4*)p);
  f32x4 v;
  if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}

// MODE 0 plain, 1 nt; every wave reads its own quarter (or WAVES-th) of the stream, UNROLL loads in flight
template <int MODE, int UNROLL>
__global__ void __launch_bounds__(1024) k_stream(const float* __restrict__ base, float* out, int kb_per_wg, int shared_stream) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float* p = base + (shared_stream ? 0 : (size_t)blockIdx.x * kb_per_wg * 256) + (size_t)wave * 256 + lane * 4;
  const int pieces = kb_per_wg / nw;   // 1 KB pieces per wave
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < pieces; i += UNROLL) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = ld<MODE>(p + (size_t)(i + u) * nw * 256);
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += v[u];
  }
  if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[blockIdx.x] = s[0];
}

// plain loads with NM independent MFMAs (constant operands, 4 accumulators) issued after every load: does a busy
// matrix pipe slow the load return path?  DEP: the MFMAs consume the loaded values of the PREVIOUS trip instead.
template <int UNROLL, int NM, bool DEP>
__global__ void __launch_bounds__(1024) k_stream_mfma(const float* __restrict__ base, float* out, int kb_per_wg, int shared_stream) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float* p = base + (shared_stream ? 0 : (size_t)blockIdx.x * kb_per_wg * 256) + (size_t)wave * 256 + lane * 4;
  const int pieces = kb_per_wg / nw;
  f32x4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float cx = 1.0f + lane * 1e-3f, cy = 0.5f;
  f32x4 prev[UNROLL];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) prev[u] = f32x4{cx, cx, cx, cx};
  for (int i = 0; i < pieces; i += UNROLL) {
    f32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      v[u] = *(gf4*)(p + (size_t)(i + u) * nw * 256);
#pragma unroll
      for (int m = 0; m < NM; ++m)
        acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(DEP ? prev[u][m & 3] : cx, cy, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) prev[u] = v[u];
  }
  f32x4 s = acc[0] + acc[1] + acc[2] + acc[3];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) s += prev[u];
  if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[blockIdx.x] = s[0];
}

// LDS-DMA: each wave streams 1 KB pieces straight into its LDS ring (no VGPR return), then reads them back
template <int UNROLL>
__global__ void __launch_bounds__(1024) k_stream_lds(const float* __restrict__ base, float* out, int kb_per_wg, int shared_stream) {
  extern __shared__ __attribute__((aligned(16))) float ring[];   // [nw][UNROLL][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const float* p = base + (shared_stream ? 0 : (size_t)blockIdx.x * kb_per_wg * 256) + (size_t)wave * 256 + lane * 4;
  const int pieces = kb_per_wg / nw;
  float* my = ring + (size_t)wave * UNROLL * 256;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)ring) + (unsigned)wave * UNROLL * 1024;   // LDS byte address of this wave's ring
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < pieces; i += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const float* src = p + (size_t)(i + u) * nw * 256;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + u * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += *(const f32x4*)(my + u * 256 + lane * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[blockIdx.x] = s[0];
}

template <typename F> float time_us(hipStream_t st, F launch) {
  for (int i = 0; i < 20; ++i) launch();
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 64;
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const size_t floats = (size_t)256 * 4096 * 256;   // 1 GB cap: nb * kb * 256 floats
  float *d, *o; CHK(hipMalloc(&d, floats * 4)); CHK(hipMalloc(&o, 4096));
  CHK(hipMemset(d, 0, floats * 4));
  CHK(hipFuncSetAttribute((const void*)k_stream_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CHK(hipFuncSetAttribute((const void*)k_stream_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  printf("%d workgroups; per-WG stream rate (GB/s and B/clk at 2.4 GHz), small/large stream differenced\n", nb);
#define RUN(NAME, KERNEL, THREADS, LDS, SHARED)                                                                    \
  do {                                                                                                            \
    const int k0 = 512, k1 = 2560;                                                                                \
    float t0 = time_us(st, [&]() { hipLaunchKernelGGL(KERNEL, dim3(nb), dim3(THREADS), LDS, st, (const float*)d, o, k0, SHARED); }); \
    float t1 = time_us(st, [&]() { hipLaunchKernelGGL(KERNEL, dim3(nb), dim3(THREADS), LDS, st, (const float*)d, o, k1, SHARED); }); \
    CHK(hipGetLastError());                                                                                       \
    const double gbs = (double)(k1 - k0) * 1024.0 / ((t1 - t0) * 1e-6) / 1e9;                                      \
    printf("  %-40s %2d waves %s  %6.1f us / %6.1f us  -> %6.1f GB/s per WG = %5.1f B/clk\n", NAME, THREADS / 64, SHARED ? "same stream " : "own stream  ", t0, t1, gbs, gbs / 2.4); \
  } while (0)
  for (int shared = 1; shared < 2; ++shared) {
    RUN("plain + 0 MFMA/load", (k_stream_mfma<8, 0, false>), 256, 0, shared);
    RUN("plain + 1 MFMA/load", (k_stream_mfma<8, 1, false>), 256, 0, shared);
    RUN("plain + 2 MFMA/load", (k_stream_mfma<8, 2, false>), 256, 0, shared);
    RUN("plain + 4 MFMA/load (=fused chain)", (k_stream_mfma<8, 4, false>), 256, 0, shared);
    RUN("plain + 8 MFMA/load", (k_stream_mfma<8, 8, false>), 256, 0, shared);
    RUN("plain + 4 MFMA/load, data-dependent", (k_stream_mfma<8, 4, true>), 256, 0, shared);
    RUN("plain + 4 MFMA/load, 8 waves", (k_stream_mfma<8, 4, false>), 512, 0, shared);
    RUN("dependent, 4 in flight", (k_stream_mfma<4, 4, true>), 256, 0, shared);
    RUN("dependent, 16 in flight", (k_stream_mfma<16, 4, true>), 256, 0, shared);
    RUN("dependent, 32 in flight", (k_stream_mfma<32, 4, true>), 256, 0, shared);
    RUN("dependent, 8 in flight, 8 waves", (k_stream_mfma<8, 4, true>), 512, 0, shared);
    RUN("dependent, 16 in flight, 8 waves", (k_stream_mfma<16, 4, true>), 512, 0, shared);
    RUN("dependent, 8 in flight, 16 waves", (k_stream_mfma<8, 4, true>), 1024, 0, shared);
  }
  for (int shared = 0; shared < 0; ++shared) {
    RUN("plain, 4 in flight/wave", (k_stream<0, 4>), 256, 0, shared);
    RUN("plain, 8 in flight/wave", (k_stream<0, 8>), 256, 0, shared);
    RUN("plain, 16 in flight/wave", (k_stream<0, 16>), 256, 0, shared);
    RUN("plain, 8 in flight/wave", (k_stream<0, 8>), 512, 0, shared);
    RUN("plain, 8 in flight/wave", (k_stream<0, 8>), 1024, 0, shared);
    RUN("plain, 4 in flight/wave", (k_stream<0, 4>), 1024, 0, shared);
    RUN("nontemporal, 8 in flight/wave", (k_stream<1, 8>), 256, 0, shared);
    RUN("nontemporal, 8 in flight/wave", (k_stream<1, 8>), 1024, 0, shared);
    RUN("LDS-DMA, 8 in flight/wave", (k_stream_lds<8>), 256, 4 * 8 * 1024, shared);
    RUN("LDS-DMA, 8 in flight/wave", (k_stream_lds<8>), 512, 8 * 8 * 1024, shared);
    RUN("LDS-DMA, 4 in flight/wave", (k_stream_lds<4>), 1024, 16 * 4 * 1024, shared);
  }
  return 0;
}
