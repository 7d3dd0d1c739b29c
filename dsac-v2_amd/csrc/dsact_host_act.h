// dsact_host_act.h -- the sampler's batch-1 acting forward ON THE HOST (SURVEY.md section 8 f1's other branch: "policy-weights
// snapshot for acting"; reference training/off_sampler.py:46-56, networks/mlp.py:79-100, utils/act_distribution_cls.py:32-42).
//
// The reference acts with a CPU copy of the whole module (ModuleOnDevice, utils/common_utils.py:164-177). The one-launch GPU
// forward of dsact_act.h costs a launch + a completion spin per ENVIRONMENT STEP (3.5 + 21 us measured, twenty times per
// iteration: 89 % of an iteration at sample_interval 1). Here the policy net (0.95 MB at Humanoid 3x256) is copied
// into pinned host memory on the handle's stream right behind every enqueued update that moves it, and dsact_act_sample /
// dsact_policy_forward(n = 1) run the forward on the host once that copy's event has fired: same semantics as the reference
// (acts with the weights of the last completed update), no launch, no spin, and environment steps that follow an update which
// leaves the policy alone overlap with it.
//
// Arithmetic: fp32, the closed forms of dsact_math.h (erf_2fit's two-interval GELU, tanh_gauss_fwd term for term); a dense
// layer is a row-major [N][K] matrix-vector product, two vector accumulators per row, rows in groups of four; the summation
// order differs from k_act_mlp's (64 lanes + a wave reduction) and from the reference's sgemv -- within the gates of
// tests/test_reference_differential.py like the GPU forward. Three instantiations of one source: 512-bit (avx512f), 256-bit
// (avx2 + fma), baseline x86-64; an output row is computed by ONE thread in ONE fixed order, so the result does not depend on
// how many threads share a layer (Pool below) -- only on the vector width the CPU offers.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__linux__)
#include <pthread.h>
#include <sched.h>
#endif

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "dsact_math.h"

namespace dsact {
namespace hostact {

struct Layer { const float* W; const float* b; int K, N; int half; };   // half > 0: a twin-trunk hidden layer -- two [half][K] blocks one
                                                                    // after the other, rows >= half read inputs [K, 2K) (dsact_act.h, ActLayer)

#if defined(__clang__)
#define DSACT_HOSTACT_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define DSACT_HOSTACT_CONTRACT
#endif

// ---- one instantiation: VB = bytes of a vector, ATTR = the function attribute that selects the instruction set.
//   exp2_neg: 2^x for x <= 0 (the only range erf_2fit's exponential sees): x = n + r, r in (-1, 0], degree-6 fit of 2^r
//             (Chebyshev-node fit in double, rounded to fp32: 1.9e-7 relative), 2^n through the exponent field; x < -126 -> 2^-126
//   erfv:     erf_2fit (dsact_math.h) on a vector, same coefficients, same interval split
//   dense:    y[n] = b[n] + sum_k W[n][k] x[k] for rows [n0, n1) (n0 a multiple of 4)
//   gelu_inplace over [i0, i1) (multiples of the lane count; the buffers are padded)
#define DSACT_HOSTACT_BODY(ATTR, VB)                                                                                              \
  typedef float vf __attribute__((vector_size(VB)));                                                                              \
  typedef int vi __attribute__((vector_size(VB)));                                                                                \
  constexpr int VL = VB / 4;                                                                                                      \
  ATTR static inline vf exp2_neg(vf x) {                                                                                          \
    const vf lo = vf{} - 126.f;                                                                                                   \
    x = x < lo ? lo : x;                                                                                                          \
    const vi n = __builtin_convertvector(x, vi);                    /* truncation: ceil for x <= 0 */                              \
    const vf r = x - __builtin_convertvector(n, vf);                                                                              \
    vf p = r * 1.0938752530e-04f + 1.2757162331e-03f;                                                                             \
    p = p * r + 9.5800580457e-03f;                                                                                                \
    p = p * r + 5.5491033942e-02f;                                                                                                \
    p = p * r + 2.4022436142e-01f;                                                                                                \
    p = p * r + 6.9314706326e-01f;                                                                                                \
    p = p * r + 1.0f;                                                                                                             \
    vi bits;                                                                                                                      \
    memcpy(&bits, &p, sizeof(bits));                                                                                              \
    bits += n << 23;                                                                                                              \
    memcpy(&p, &bits, sizeof(p));                                                                                                 \
    return p;                                                                                                                     \
  }                                                                                                                               \
  ATTR static inline vf erfv(vf x) {                                                                                              \
    const vf ax = x < 0.f ? -x : x;                                                                                               \
    const vf hi = vf{} + kErfHi;                                                                                                  \
    const vf t = ax > hi ? hi : ax;                                                                                               \
    const vf s = x * x;                                                                                                           \
    vf rs = s * kErfS[5] + kErfS[4];                                                                                              \
    rs = rs * s + kErfS[3]; rs = rs * s + kErfS[2]; rs = rs * s + kErfS[1]; rs = rs * s + kErfS[0];                                \
    const vf small = rs * x + x;                                                                                                  \
    vf rl = t * kErfL[8] + kErfL[7];                                                                                              \
    rl = rl * t + kErfL[6]; rl = rl * t + kErfL[5]; rl = rl * t + kErfL[4]; rl = rl * t + kErfL[3];                                \
    rl = rl * t + kErfL[2]; rl = rl * t + kErfL[1]; rl = rl * t + kErfL[0];                                                        \
    const vf e = exp2_neg((rl * t - t) * kLog2e);                                                                                 \
    vf large = 1.0f - e;                                                                                                          \
    large = x < 0.f ? -large : large;                                                                                             \
    return t > kErfT0 ? large : small;                                                                                            \
  }                                                                                                                               \
  ATTR static inline float hsum(vf a) {                                                                                           \
    float s[VL];                                                                                                                  \
    memcpy(s, &a, sizeof(s));                                                                                                     \
    for (int w = VL / 2; w >= 1; w /= 2)                                                                                          \
      for (int i = 0; i < w; ++i) s[i] += s[i + w];                                                                               \
    return s[0];                                                                                                                  \
  }                                                                                                                               \
  ATTR static void dense(const Layer& L, const float* x, float* y, int n0, int n1) {                                              \
    DSACT_HOSTACT_CONTRACT                                                                                                        \
    const int K = L.K, KV = K - K % (2 * VL);                                                                                     \
    int n = n0;                                                                                                                   \
    for (; n + 4 <= n1; n += 4) {                                                                                                 \
      const float* w0 = L.W + (size_t)n * K;                                                                                      \
      const float *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;                                                                       \
      vf a0 = vf{}, a1 = a0, a2 = a0, a3 = a0, c0 = a0, c1 = a0, c2 = a0, c3 = a0;                                                \
      for (int k = 0; k < KV; k += 2 * VL) {                                                                                      \
        vf xa, xb, t;                                                                                                             \
        memcpy(&xa, x + k, VB); memcpy(&xb, x + k + VL, VB);                                                                      \
        memcpy(&t, w0 + k, VB); a0 += t * xa; memcpy(&t, w0 + k + VL, VB); c0 += t * xb;                                           \
        memcpy(&t, w1 + k, VB); a1 += t * xa; memcpy(&t, w1 + k + VL, VB); c1 += t * xb;                                           \
        memcpy(&t, w2 + k, VB); a2 += t * xa; memcpy(&t, w2 + k + VL, VB); c2 += t * xb;                                           \
        memcpy(&t, w3 + k, VB); a3 += t * xa; memcpy(&t, w3 + k + VL, VB); c3 += t * xb;                                           \
      }                                                                                                                           \
      float s0 = hsum(a0 + c0), s1 = hsum(a1 + c1), s2 = hsum(a2 + c2), s3 = hsum(a3 + c3);                                       \
      for (int k = KV; k < K; ++k) { const float xv = x[k]; s0 += w0[k] * xv; s1 += w1[k] * xv; s2 += w2[k] * xv; s3 += w3[k] * xv; } \
      y[n] = s0 + L.b[n]; y[n + 1] = s1 + L.b[n + 1]; y[n + 2] = s2 + L.b[n + 2]; y[n + 3] = s3 + L.b[n + 3];                     \
    }                                                                                                                             \
    for (; n < n1; ++n) {                                                                                                         \
      const float* w0 = L.W + (size_t)n * K;                                                                                      \
      vf a0 = vf{}, c0 = a0;                                                                                                      \
      for (int k = 0; k < KV; k += 2 * VL) {                                                                                      \
        vf xa, xb, t;                                                                                                             \
        memcpy(&xa, x + k, VB); memcpy(&xb, x + k + VL, VB);                                                                      \
        memcpy(&t, w0 + k, VB); a0 += t * xa; memcpy(&t, w0 + k + VL, VB); c0 += t * xb;                                           \
      }                                                                                                                           \
      float s0 = hsum(a0 + c0);                                                                                                   \
      for (int k = KV; k < K; ++k) s0 += w0[k] * x[k];                                                                            \
      y[n] = s0 + L.b[n];                                                                                                         \
    }                                                                                                                             \
  }                                                                                                                               \
  ATTR static void gelu_inplace(float* y, int i0, int i1) {                                                                       \
    DSACT_HOSTACT_CONTRACT                                                                                                        \
    for (int i = i0; i < i1; i += VL) {                                                                                           \
      vf z;                                                                                                                       \
      memcpy(&z, y + i, VB);                                                                                                      \
      const vf cdf = erfv(z * kInvSqrt2) * 0.5f + 0.5f;                                                                           \
      z = z * cdf;                                                                                                                \
      memcpy(y + i, &z, VB);                                                                                                      \
    }                                                                                                                             \
  }                                                                                                                               \
  /* rows [n0, n1) of a layer (n0, n1 multiples of 16 or the layer's end) + its hidden activation */                             \
  ATTR static void layer_rows(const Layer& L, int act, bool hidden, const float* x, float* y, int n0, int n1) {                   \
    if (L.half > 0) {                                                                                                             \
      const int m = n1 < L.half ? n1 : L.half, q = n0 > L.half ? n0 : L.half;                                                     \
      if (n0 < m) dense(L, x, y, n0, m);                                                                                          \
      if (q < n1) dense(L, x + L.K, y, q, n1);                                                                                    \
    } else dense(L, x, y, n0, n1);                                                                                                \
    if (!hidden) return;                                                                                                          \
    const int np = (n1 + 15) & ~15;                     /* (the last chunk also owns the padding behind the layer's end) */       \
    for (int i = n1; i < np; ++i) y[i] = 0.f;                                                                                     \
    if (act == ACT_GELU) gelu_inplace(y, n0, np);                                                                                 \
    else for (int i = n0; i < n1; ++i) { float hv, gd; act_fwd_grad(act, y[i], hv, gd); y[i] = hv; }                              \
    for (int i = n1; i < np; ++i) y[i] = 0.f;                                                                                     \
  }

namespace avx512 {
DSACT_HOSTACT_BODY(__attribute__((target("avx512f,fma"))), 64)
}  // namespace avx512
namespace avx2 {
DSACT_HOSTACT_BODY(__attribute__((target("avx2,fma"))), 32)
}  // namespace avx2
namespace base {
DSACT_HOSTACT_BODY(, 32)
}  // namespace base
#undef DSACT_HOSTACT_BODY

enum : int { ISA_BASE = 0, ISA_AVX2 = 1, ISA_AVX512 = 2 };
inline int cpu_isa() {
  static const int isa = (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("fma")) ? ISA_AVX512
                         : (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? ISA_AVX2 : ISA_BASE;
  return isa;
}
inline bool cpu_has_avx2_fma() { return cpu_isa() >= ISA_AVX2; }
inline void layer_rows(int isa, const Layer& L, int act, bool hidden, const float* x, float* y, int n0, int n1) {
  if (isa == ISA_AVX512) avx512::layer_rows(L, act, hidden, x, y, n0, n1);
  else if (isa == ISA_AVX2) avx2::layer_rows(L, act, hidden, x, y, n0, n1);
  else base::layer_rows(L, act, hidden, x, y, n0, n1);
}

// ---- a small fork-join pool for the wide layers: the caller is worker 0, T - 1 helpers spin for the next layer while a burst
// of acting calls lasts (an environment step between two calls is microseconds) and go to sleep on a condition variable when
// none has come for ~200 us. A layer's rows are dealt in contiguous chunks of multiples of 16; chunk boundaries do not change
// any row's arithmetic. Not created at all for T == 1 (DSACT_HOST_ACT_THREADS=1).
// CPUs that share the last-level cache with `cpu` (Linux sysfs), one per physical core, `cpu`'s own core excluded: where the
// helpers of a Pool want to run -- a layer's fork-join is two cache-line hand-overs, which cost tens of nanoseconds inside a
// core complex and a microsecond across sockets. Empty: unknown topology (the helpers then run wherever the scheduler puts them).
inline std::vector<int> llc_sibling_cores(int cpu) {
  std::vector<int> out;
#if defined(__linux__)
  auto read_list = [](const char* path, std::vector<int>& v) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    char buf[4096];
    const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
    fclose(f);
    if (!ok) return false;
    for (char* p = buf; *p && *p != '\n';) {        // "0-7,128-135"
      char* e;
      const long a = strtol(p, &e, 10);
      if (e == p) break;
      long b = a;
      if (*e == '-') { p = e + 1; b = strtol(p, &e, 10); }
      for (long c = a; c <= b && c - a < 4096; ++c) v.push_back((int)c);
      p = *e == ',' ? e + 1 : e;
    }
    return !v.empty();
  };
  char path[160];
  std::vector<int> llc, mine;
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
  if (!read_list(path, llc)) return out;
  snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
  read_list(path, mine);
  std::vector<int> taken = mine;                     // hardware threads of cores already used
  for (int c : llc) {
    bool used = false;
    for (int t : taken) used = used || t == c;
    if (used || c == cpu) continue;
    std::vector<int> sib;
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    if (!read_list(path, sib)) sib.push_back(c);
    for (int t : sib) taken.push_back(t);
    out.push_back(c);
  }
#endif
  return out;
}

class Pool {
 public:
  // pin_to: CPU of helper i (i = 1 .. threads-1) at pin_to[i - 1], or fewer entries / empty: unpinned helpers
  explicit Pool(int threads, const std::vector<int>& pin_to = std::vector<int>()) : T_(threads < 1 ? 1 : threads) {
    for (int i = 1; i < T_; ++i) {
      th_.emplace_back([this, i] { run(i); });
#if defined(__linux__)
      if ((size_t)(i - 1) < pin_to.size()) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(pin_to[(size_t)i - 1], &set);
        if (pthread_setaffinity_np(th_.back().native_handle(), sizeof(set), &set) == 0) ++pinned_;
      }
#endif
    }
  }
  int pinned() const { return pinned_; }
  // move the helpers (the calling thread migrated to another core complex: see llc_sibling_cores); returns the number pinned
  int repin(const std::vector<int>& pin_to) {
    pinned_ = 0;
#if defined(__linux__)
    for (size_t i = 0; i < th_.size() && i < pin_to.size(); ++i) {
      cpu_set_t set;
      CPU_ZERO(&set);
      CPU_SET(pin_to[i], &set);
      if (pthread_setaffinity_np(th_[i].native_handle(), sizeof(set), &set) == 0) ++pinned_;
    }
#endif
    return pinned_;
  }
  ~Pool() {
    stop_.store(true);
    gen_.fetch_add(1);
    { std::lock_guard<std::mutex> g(m_); }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int threads() const { return T_; }
  // rows [0, N) of layer L on all workers; returns when every row is written
  void layer(int isa, const Layer& L, int act, bool hidden, const float* x, float* y) {
    const int per = (((L.N + T_ - 1) / T_) + 15) & ~15;
    if (T_ == 1 || L.N < 64 || per >= L.N) { layer_rows(isa, L, act, hidden, x, y, 0, L.N); return; }
    job_ = Job{isa, &L, act, hidden, x, y, per};
    done_.store(0, std::memory_order_relaxed);
    gen_.fetch_add(1);                                   // (seq_cst: publishes job_, orders against sleepers_ below)
    if (sleepers_.load() > 0) { { std::lock_guard<std::mutex> g(m_); } cv_.notify_all(); }
    slice(0);
    while (done_.load(std::memory_order_acquire) < T_ - 1) __builtin_ia32_pause();
  }

 private:
  struct Job { int isa; const Layer* L; int act; bool hidden; const float* x; float* y; int per; };
  void slice(int id) {
    const Job j = job_;
    const int n0 = id * j.per, n1 = n0 + j.per < j.L->N ? n0 + j.per : j.L->N;
    if (n0 < n1) layer_rows(j.isa, *j.L, j.act, j.hidden, j.x, j.y, n0, n1);
  }
  void run(int id) {
    unsigned seen = 0;
    for (;;) {
      int spins = 0;
      while (gen_.load(std::memory_order_acquire) == seen) {
        __builtin_ia32_pause();
        if (++spins > 20000) {                           // ~200 us without work: sleep until the next burst
          std::unique_lock<std::mutex> lk(m_);
          sleepers_.fetch_add(1);
          cv_.wait(lk, [&] { return gen_.load() != seen || stop_.load(); });
          sleepers_.fetch_sub(1);
          spins = 0;
        }
      }
      seen = gen_.load(std::memory_order_acquire);
      if (stop_.load()) return;
      slice(id);
      done_.fetch_add(1, std::memory_order_release);
    }
  }
  const int T_;
  int pinned_ = 0;
  std::vector<std::thread> th_;
  std::atomic<unsigned> gen_{0};
  std::atomic<int> done_{0}, sleepers_{0};
  std::atomic<bool> stop_{false};
  std::mutex m_;
  std::condition_variable cv_;
  Job job_{};
};

// hidden layers + the output layer's raw products: out[0 .. N_last). buf0 / buf1: activation buffers of >= widest layer + 64
// floats. pool == nullptr: everything on the calling thread.
inline void forward(const Layer* ly, int n_layers, int act, const float* obs, float* buf0, float* buf1, float* out, Pool* pool = nullptr,
                    int isa = -1) {
  if (isa < 0) isa = cpu_isa();
  const float* x = obs;
  float* nxt = buf0;
  for (int l = 0; l < n_layers; ++l) {
    const bool hidden = l + 1 < n_layers;
    float* y = hidden ? nxt : out;
    if (pool) pool->layer(isa, ly[l], act, hidden, x, y);
    else layer_rows(isa, ly[l], act, hidden, x, y, 0, ly[l].N);
    x = y;
    nxt = nxt == buf0 ? buf1 : buf0;
  }
}

// dsact_act_sample on the host: policy(obs) + TanhGaussDistribution.sample() (utils/act_distribution_cls.py:32-42) with the
// caller's N(0,1) draw -- tanh_gauss_fwd of dsact_math.h term for term, as k_act_mlp's sample mode. raw: the output layer's
// 2A products (mean | raw log-std). eps == nullptr: out = the 2A logits (mean | exp(clamp(raw))) of StochaPolicy.forward
// (networks/mlp.py:85-100); else action[A] and the summed log-probability.
inline void head(const float* raw_in, int A, float lo_ls, float hi_ls, const float* eps, const float* scale, const float* center,
                 float* action_or_logits, float* logp, int out_act = 0, int out_n = 0) {
  float act_buf[64];
  const float* raw = raw_in;
  if (out_act) {   // policy_output_activation: the module that follows the last Linear, on the outputs it applies to
    for (int i = 0; i < 2 * A; ++i) act_buf[i] = i < out_n ? out_act_fwd(out_act, raw_in[i]) : raw_in[i];
    raw = act_buf;
  }
  if (!eps) {
    for (int d = 0; d < A; ++d) { action_or_logits[d] = raw[d]; action_or_logits[A + d] = expf(clampf(raw[A + d], lo_ls, hi_ls)); }
    return;
  }
  float lp = 0.0f;
  for (int d = 0; d < A; ++d) {
    const TanhGaussFwd f = tanh_gauss_fwd(raw[d], raw[A + d], eps[d], scale[d], center[d], lo_ls, hi_ls);
    action_or_logits[d] = f.a;
    lp += f.lp;            // Independent(..., 1): sum over the action dimensions, in order
  }
  *logp = lp;
}

}  // namespace hostact
}  // namespace dsact
