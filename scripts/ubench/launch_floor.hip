// per-kernel cost of a chain of dependent kernels replayed from a hipGraph (same stream), for a few
// kernel shapes: empty, 256 workgroups touching 64 KB each of L2-resident data, with/without 73 KB of LDS
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_empty(float* p) { if (p == nullptr) p[0] = 1.f; }
__global__ void __launch_bounds__(256) k_touch(const float* __restrict__ in, float* __restrict__ out) {
  extern __shared__ float lds[];
  f32x4 v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = *(const f32x4*)(in + ((size_t)(blockIdx.x & 63) * 16 + j) * 1024 + threadIdx.x * 4);
  f32x4 s = v[0];
#pragma unroll
  for (int j = 1; j < 16; ++j) s += v[j];
  lds[threadIdx.x] = s.x;
  __syncthreads();
  *(f32x4*)(out + (size_t)blockIdx.x * 1024 + threadIdx.x * 4) = s + lds[(threadIdx.x + 1) & 255];
}
template <typename F> float run_graph(hipStream_t st, int n, F enqueue) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) enqueue(i);
  hipStreamEndCapture(st, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st);
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  const int reps = 200;
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
  hipEventRecord(e1, st);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return ms * 1000.f / (reps * n);
}
int main() {
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *a, *b; CHK(hipMalloc(&a, 64 * 16 * 1024 * 4)); CHK(hipMalloc(&b, 256 * 1024 * 4 * 2));
  CHK(hipMemset(a, 0, 64 * 16 * 1024 * 4));
  CHK(hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  const int n = 16;
  printf("us per kernel in a %d-kernel dependent chain (hipGraph replay):\n", n);
  printf("  empty, 1 block          : %.2f\n", run_graph(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, a); }));
  printf("  empty, 256 blocks       : %.2f\n", run_graph(st, n, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, a); }));
  printf("  touch 64KB/WG, 1KB LDS  : %.2f\n", run_graph(st, n, [&](int i) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 1024, st, (const float*)a, b + (i & 1) * 256 * 1024); }));
  printf("  touch 64KB/WG, 73KB LDS : %.2f\n", run_graph(st, n, [&](int i) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 73728, st, (const float*)a, b + (i & 1) * 256 * 1024); }));
  printf("  touch 64KB/WG, 129KB LDS: %.2f\n", run_graph(st, n, [&](int i) { hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 129024, st, (const float*)a, b + (i & 1) * 256 * 1024); }));
  return 0;
}
