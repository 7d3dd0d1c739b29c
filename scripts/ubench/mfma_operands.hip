// Does v_mfma_f32_16x16x4_f32 issue at 32 cycles when its A/B operands come from DIFFERENT VGPRs every time?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
// NA: distinct A-operand registers cycled through; NB: same for B; NACC accumulators
template <int NA, int NB, int NACC>
__global__ void __launch_bounds__(256) k(long long* out, int iters, float seed) {
  float a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = seed * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = seed * (threadIdx.x * 3 + i);
#pragma unroll
  for (int i = 0; i < NA; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
  for (int i = 0; i < NB; ++i) asm volatile("" : "+v"(b[i]));
  f32x4 acc[NACC];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const long long c0 = (long long)__builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 32; ++m) acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m % NA], b[m % NB], acc[m % NACC], 0, 0, 0);
  }
  const long long c1 = (long long)__builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int q = 1; q < NACC; ++q) s += acc[q];
  if (threadIdx.x == 0) out[blockIdx.x] = c1 - c0;
  if (s[0] + s[1] == 1.2345f) out[0] = 0;
}
int main() {
  long long* d; CHK(hipMalloc(&d, 8 * 256));
  long long h[256];
#define RUN(NA, NB, NACC) do { hipLaunchKernelGGL((k<NA, NB, NACC>), dim3(8), dim3(256), 0, 0, d, 2000, 1e-3f); CHK(hipDeviceSynchronize()); \
    CHK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost)); printf("A regs %2d  B regs %2d  acc %d : %.1f cycles per MFMA\n", NA, NB, NACC, (double)h[0] / (2000.0 * 32)); } while (0)
  RUN(1, 1, 4); RUN(4, 1, 4); RUN(16, 1, 4); RUN(1, 16, 4); RUN(16, 16, 4); RUN(32, 8, 4); RUN(32, 32, 4); RUN(16, 16, 2); RUN(16, 16, 8); RUN(32, 2, 4);
  return 0;
}
