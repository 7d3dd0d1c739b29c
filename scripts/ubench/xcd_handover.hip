// Latency of an in-launch hand-over between two CUs of the SAME XCD vs two CUs on DIFFERENT XCDs (VERDICT r4, item 1
// probe): can a slice-layer of the DSAC-T chains be split across CUs with an exchange cheaper than the ~1.5 us a
// tagged agent-scope hand-over costs inside k_chain_fwdp?
//
// Two workgroups (256 threads, one per CU: 100 KB of LDS each) ping-pong; everything else in the grid exits.
//   mode 0  granule : one naturally aligned 8-byte {value, tag} pair, producer store / consumer poll with the given
//                     scope bits -- the chains' tagged hand-over (csrc/dsact_chain.h, HW_PAIRS_*)
//   mode 1  4 KB    : 256 lanes x 16 B payload (stores with the producer's scope bits), every wave drains
//                     (s_waitcnt vmcnt(0)), barrier, lane 0 raises an 8-byte flag; the consumer polls the flag, then all
//                     lanes load the payload (16 B each) and CHECK it -- an activation exchange of 8 rows x 128 features
//   mode 2  4 KB tagged: 512 granules (2 per lane), no flag: every lane polls its own pairs (2 KB of payload per 4 KB)
// store scopes: "plain" (stays in the XCD's L2), "sc0", "sc1" (write-through, what the chains use), "sc0sc1";
// load scopes: "sc1" (bypasses the CU's L1, served by L2 / memory), "sc0sc1". (sc0 / plain loads may hit the reader's
// L1 for ever: not a hand-over; the bounded spin reports them as timeouts when asked for.)
// load = 1: both endpoint workgroups keep a weight stream in flight (each wave 16 x 1 KB loads per round from a 4 MB
// region, the k_chain_* geometry) between the hand-overs -- the condition inside the chain kernels.
// Output: one-way latency = round trip / 2, median of 5 runs of 2000 round trips; stale = payload words that did not
// carry the expected value (validity of the scope combination at that placement).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

enum { ST_PLAIN = 0, ST_SC0 = 1, ST_SC1 = 2, ST_SC01 = 3 };
enum { LD_SC1 = 0, LD_SC01 = 1, LD_SC0 = 2 };

template <int S> __device__ __forceinline__ void st8(u64* p, u64 v) {
  if (S == ST_PLAIN) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else if (S == ST_SC0) asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  else if (S == ST_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
template <int S> __device__ __forceinline__ void st16(float* p, f32x4 v) {
  if (S == ST_PLAIN) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else if (S == ST_SC0) asm volatile("global_store_dwordx4 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  else if (S == ST_SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}
template <int S> __device__ __forceinline__ u64 ld8(const u64* p) {
  u64 v;
  if (S == LD_SC1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (S == LD_SC01) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int S> __device__ __forceinline__ f32x4 ld16(const float* p) {
  f32x4 v;
  if (S == LD_SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (S == LD_SC01) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Args {
  u64* flag[2];        // [0]: written by A, polled by B; [1]: the way back
  float* pay[2];       // 4 KB payload each way
  u64* pairs[2];       // 512 granules each way
  const float* wts;    // 4 MB "weights" (the background stream)
  int blk_a, blk_b, iters, mode, load;
  long long* out;      // [0] ticks (100 MHz), [1] stale words, [2] timeouts, [3]/[4] XCC_ID of A / B, [5]/[6] HW_ID
  float* sink;
};

constexpr int kSpin = 1 << 16;

template <int ST, int LD>
__global__ void __launch_bounds__(256) k_pingpong(Args a) {
  extern __shared__ float lds[];
  const int b = (int)blockIdx.x;
  if (b != a.blk_a && b != a.blk_b) return;
  const int me = b == a.blk_a ? 0 : 1, other = me ^ 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    a.out[3 + me] = __builtin_amdgcn_s_getreg((3 << 11) | 20);    // HW_REG_XCC_ID[3:0]
    a.out[5 + me] = __builtin_amdgcn_s_getreg((15 << 11) | 4);    // HW_REG_HW_ID[15:0]
  }
  long long stale = 0, timeouts = 0;
  bool give_up = false;
  f32x4 wacc = {0.f, 0.f, 0.f, 0.f};
  const float* wbase = a.wts + (size_t)wave * 256 * 1024 + lane * 4;   // 1 MB per wave, 1 KB wave-loads
  int wpos = 0;
  __syncthreads();
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  for (int i = 1; i <= a.iters; ++i) {
    const u64 tag = (u64)(unsigned)i;
    for (int half = 0; half < 2; ++half) {
      const bool sender = (half == 0) == (me == 0);
      f32x4 wbuf[16];
      if (a.load) {   // 16 KB per wave in flight ACROSS the hand-over (consumed after it): the polls queue behind them
#pragma unroll
        for (int u = 0; u < 16; ++u) wbuf[u] = *(const f32x4*)(wbase + (size_t)((wpos + u) & 1023) * 256);
        wpos += 16;
      }
      if (sender) {
        if (a.mode == 0) {
          if (tid == 0) st8<ST>(a.flag[me], (tag << 32) | tag);
        } else if (a.mode == 1) {
          f32x4 v = {(float)i, (float)i, (float)i, (float)i};
          st16<ST>(a.pay[me] + tid * 4, v);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (tid == 0) st8<ST_SC1>(a.flag[me], (tag << 32) | tag);
        } else {
          st8<ST>(a.pairs[me] + tid, (tag << 32) | tag);
          st8<ST>(a.pairs[me] + 256 + tid, (tag << 32) | tag);
        }
      } else {
        if (a.mode == 0 || a.mode == 1) {
          if (tid == 0) {
            int spins = 0;
            u64 v = a.mode == 0 ? ld8<LD>(a.flag[other]) : ld8<LD_SC1>(a.flag[other]);
            while ((v >> 32) != tag) {
              if (++spins > kSpin) { ++timeouts; break; }
              v = a.mode == 0 ? ld8<LD>(a.flag[other]) : ld8<LD_SC1>(a.flag[other]);
            }
          }
          __syncthreads();
          if (a.mode == 1) {
            const f32x4 v = ld16<LD>(a.pay[other] + tid * 4);
            stale += (v[0] != (float)i) + (v[1] != (float)i) + (v[2] != (float)i) + (v[3] != (float)i);
          }
        } else {
          for (int q = 0; q < 2; ++q) {
            int spins = 0;
            u64 v = ld8<LD>(a.pairs[other] + q * 256 + tid);
            while ((v >> 32) != tag) {
              if (++spins > kSpin) { ++timeouts; break; }
              v = ld8<LD>(a.pairs[other] + q * 256 + tid);
            }
          }
          __syncthreads();
        }
      }
      if (a.load) {
#pragma unroll
        for (int u = 0; u < 16; ++u) wacc += wbuf[u];
      }
      if (__syncthreads_or(timeouts != 0)) { give_up = true; break; }   // workgroup-uniform; the partner times out in turn
    }
    if (give_up) break;
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memrealtime();
  a.sink[b * 256 + tid] = wacc[0] + wacc[1] + wacc[2] + wacc[3];
  atomicAdd((unsigned long long*)&a.out[1], (unsigned long long)stale);
  atomicAdd((unsigned long long*)&a.out[2], (unsigned long long)timeouts);
  if (me == 0 && tid == 0) a.out[0] = t1 - t0;
}

typedef void (*Kern)(Args);
struct Variant { const char* name; Kern k; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  Args a;
  memset(&a, 0, sizeof(a));
  char* buf; CHK(hipMalloc(&buf, 1 << 20)); CHK(hipMemset(buf, 0, 1 << 20));
  a.flag[0] = (u64*)buf; a.flag[1] = (u64*)(buf + 4096);
  a.pay[0] = (float*)(buf + 8192); a.pay[1] = (float*)(buf + 16384);
  a.pairs[0] = (u64*)(buf + 32768); a.pairs[1] = (u64*)(buf + 65536);
  float* w; CHK(hipMalloc(&w, 4 << 20)); CHK(hipMemset(w, 0, 4 << 20));
  a.wts = w;
  long long* out; CHK(hipMalloc(&out, 64)); a.out = out;
  float* sink; CHK(hipMalloc(&sink, 64 * 256 * 4)); a.sink = sink;
  a.iters = iters;
  const Variant V[] = {
    {"st sc1    / ld sc1   ", k_pingpong<ST_SC1, LD_SC1>},
    {"st plain  / ld sc1   ", k_pingpong<ST_PLAIN, LD_SC1>},
    {"st sc0    / ld sc1   ", k_pingpong<ST_SC0, LD_SC1>},
    {"st sc0sc1 / ld sc0sc1", k_pingpong<ST_SC01, LD_SC01>},
    {"st sc0    / ld sc0   ", k_pingpong<ST_SC0, LD_SC0>},
  };
  const char* modes[] = {"8-byte granule", "4 KB payload + flag", "4 KB of tagged granules (2 KB payload)"};
  // block b runs on XCD b % 8 (checked below through XCC_ID): (0, 8) = same XCD, (0, 1) = neighbours, (0, 4) = far
  const int pairs[][2] = {{0, 8}, {0, 1}, {0, 4}};
  const char* pname[] = {"same XCD (blocks 0, 8)", "cross XCD (blocks 0, 1)", "cross XCD (blocks 0, 4)"};
  for (const Variant& v : V) CHK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  printf("one-way hand-over latency between two CUs, us (round trip / 2, median of 5 x %d round trips)\n", iters);
  for (int load = 0; load < 2; ++load) {
    printf("\n== endpoint CUs %s\n", load ? "STREAMING (16 x 1 KB wave-loads per wave in flight, as in the chain kernels)" : "idle");
    for (int mode = 0; mode < 3; ++mode) {
      printf("-- %s\n", modes[mode]);
      for (const Variant& v : V) {
        for (int p = 0; p < 3; ++p) {
          a.blk_a = pairs[p][0]; a.blk_b = pairs[p][1]; a.mode = mode; a.load = load;
          double us[5]; long long o[8] = {0};
          for (int rep = 0; rep < 5; ++rep) {
            CHK(hipMemsetAsync(buf, 0, 1 << 20, st));
            CHK(hipMemsetAsync(out, 0, 64, st));
            hipLaunchKernelGGL(v.k, dim3(16), dim3(256), 100 * 1024, st, a);
            CHK(hipStreamSynchronize(st));
            CHK(hipMemcpy(o, out, 64, hipMemcpyDeviceToHost));
            us[rep] = o[2] ? -1.0 : (double)o[0] * 0.01 / (2.0 * iters);
          }
          std::sort(us, us + 5);
          printf("   %s  %-26s  %7.3f us   stale %lld  timeouts %lld   xcc %lld/%lld  cu %lld/%lld\n", v.name, pname[p], us[2], o[1], o[2],
                 o[3], o[4], (o[5] >> 8) & 15, (o[6] >> 8) & 15);
        }
      }
    }
  }
  return 0;
}
