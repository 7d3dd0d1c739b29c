// TEST INFRASTRUCTURE ONLY: the scalar closed forms of dsac-v2_amd/csrc/dsact_math.h compiled for
// the host so that tests/test_host_math.py can check them against torch autograd on a CPU box.
// Never loaded by the product.
#include "dsact_math.h"

extern "C" {
void hm_gelu(const float* z, int n, float* h, float* g) {
  for (int i = 0; i < n; ++i) dsact::gelu_fwd_grad(z[i], h[i], g[i]);
}
void hm_softplus(const float* x, int n, float* y, float* dy) {
  for (int i = 0; i < n; ++i) { y[i] = dsact::softplus(x[i]); dy[i] = dsact::softplus_grad(x[i]); }
}
void hm_tanh_gauss_fwd(const float* mu, const float* raw, const float* eps, int n, float s, float c, float lo,
                       float hi, float* a, float* lp) {
  for (int i = 0; i < n; ++i) {
    dsact::TanhGaussFwd f = dsact::tanh_gauss_fwd(mu[i], raw[i], eps[i], s, c, lo, hi);
    a[i] = f.a; lp[i] = f.lp;
  }
}
void hm_tanh_gauss_bwd(const float* mu, const float* raw, const float* eps, const float* gA, int n, float s,
                       float lo, float hi, float gLp, float* dmu, float* draw) {
  for (int i = 0; i < n; ++i) dsact::tanh_gauss_bwd(mu[i], raw[i], eps[i], s, lo, hi, gA[i], gLp, dmu[i], draw[i]);
}
void hm_critic(const float* q, const float* stdv, const float* tq, const float* tqs, int n, float ms, float* loss,
               float* dq, float* dstd) {
  for (int i = 0; i < n; ++i) {
    dsact::CriticTerm t = dsact::critic_term(q[i], stdv[i], ms, tq[i], tqs[i]);
    loss[i] = t.loss; dq[i] = t.dq; dstd[i] = t.dstd;
  }
}
void hm_adam(float* p, float* m, float* v, const float* g, int n, float b1w, float beta2, float b2w, float ss,
             float bc2, float eps) {
  for (int i = 0; i < n; ++i) dsact::adam_update(p[i], m[i], v[i], g[i], b1w, beta2, b2w, ss, bc2, eps);
}
void hm_polyak(float* pt, const float* p, int n, float polyak, float one_minus) {
  for (int i = 0; i < n; ++i) pt[i] = dsact::polyak_update(pt[i], p[i], polyak, one_minus);
}
void hm_out_act(int act, float z, float* y, float* dy) {   // output activations: y = act(z), dy = d act / dz expressed through y
  *y = dsact::out_act_fwd(act, z);
  *dy = dsact::out_act_grad(act, *y, z);   // (== out_act_grad_y(act, y) for every activation but OUT_ACT_GELU)
}
}

// the host-side acting forward of the product (dsac-v2_amd/csrc/dsact_host_act.h: the sampler's batch-1 policy forward on the
// calling thread) over a flat parameter vector in arena order, for tests/test_host_math.py
#include "dsact_host_act.h"
extern "C" {
int hm_policy_act(const float* params, int n_layers, const int* k_in, const int* n_out, const long long* w_off,
                  const long long* b_off, int act, const float* obs, int A, float lo_ls, float hi_ls, const float* eps,
                  const float* scale, const float* center, float* out, float* logp, int isa, int threads, const int* half) {
  // half (may be NULL): per layer, > 0 = a twin-trunk hidden layer of two [half][k_in] blocks (policy_std_type "mlp_separated")
  // isa: -1 = what the CPU offers, 0 baseline x86-64, 1 avx2 + fma, 2 avx512f (refused with -2 when the CPU lacks it)
  dsact::hostact::Layer ly[8];
  if (n_layers > 8) return -1;
  if (isa > dsact::hostact::cpu_isa()) return -2;
  int widest = 0;
  for (int l = 0; l < n_layers; ++l) {
    ly[l].W = params + w_off[l]; ly[l].b = params + b_off[l]; ly[l].K = k_in[l]; ly[l].N = n_out[l];
    ly[l].half = half ? half[l] : 0;
    if (n_out[l] > widest) widest = n_out[l];
  }
  float* buf = new float[3 * (widest + 64)];
  float* raw = buf + 2 * (widest + 64);
  {
    dsact::hostact::Pool pool(threads);
    dsact::hostact::forward(ly, n_layers, act, obs, buf, buf + widest + 64, raw, threads > 1 ? &pool : nullptr, isa);
  }
  dsact::hostact::head(raw, A, lo_ls, hi_ls, eps, scale, center, out, logp);
  delete[] buf;
  return dsact::hostact::cpu_isa();
}
}
