// dsact_math.h -- scalar fp32 math of the DSAC-T update, shared by every kernel.
// Host+device so that the closed forms can be checked on a CPU box (tests/test_host_math.py builds
// them with g++ into a tiny test-only library); the product only ever runs them on gfx950.
//
// Reference semantics (file:line relative to Jingliang-Duan/DSAC-v2):
//   GELU (exact erf)           utils/common_utils.py:25-26 -> torch.nn.GELU()
//   softplus(beta=1,thr=20)    networks/mlp.py:125
//   tanh-Gaussian rsample      utils/act_distribution_cls.py:44-54
//   critic target / loss       dsac_v2.py:218-302
//   Adam (single tensor)       torch.optim.Adam defaults, dsac_v2.py:54-59,321-328
//   Polyak                     dsac_v2.py:330-347
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DSACT_HD __host__ __device__ __forceinline__
#else
#define DSACT_HD inline
#endif

namespace dsact {

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;
constexpr float kLogSqrt2Pi = 0.91893853320467274178f;  // math.log(math.sqrt(2*math.pi))
constexpr float kTanhEps = 1e-6f;                        // act_distribution_cls.py:3

// gelu(z) and d gelu / dz share the erf; both are stored by the forward epilogue.
DSACT_HD void gelu_fwd_grad(float z, float& h, float& g) {
  const float cdf = 0.5f * (1.0f + erff(z * kInvSqrt2));
  const float pdf = kInvSqrt2Pi * expf(-0.5f * z * z);
  h = z * cdf;
  g = cdf + z * pdf;
}

DSACT_HD float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }
// d softplus / dx (torch: grad * (x*beta > threshold ? 1 : z/(z+1)), z = exp(x))
DSACT_HD float softplus_grad(float x) {
  if (x > 20.0f) return 1.0f;
  const float e = expf(x);
  return e / (e + 1.0f);
}

DSACT_HD float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// ---- tanh-Gaussian rsample, one action dimension ------------------------------------------------
// mu, raw: policy head outputs; eps ~ N(0,1); s=(hi-lo)/2, c=(hi+lo)/2.
// returns action a and this dimension's log-prob contribution lp.
struct TanhGaussFwd {
  float a, lp, sigma, t;
};
DSACT_HD TanhGaussFwd tanh_gauss_fwd(float mu, float raw, float eps, float s, float c, float lo_ls,
                                     float hi_ls) {
  TanhGaussFwd o;
  o.sigma = expf(clampf(raw, lo_ls, hi_ls));
  const float x = mu + eps * o.sigma;  // Normal.rsample: loc + eps*scale
  o.t = tanhf(x);
  o.a = s * o.t + c;
  const float d = x - mu;
  const float var = o.sigma * o.sigma;
  // Normal.log_prob: -((v-loc)^2)/(2 var) - log(scale) - log(sqrt(2 pi))
  float lp = -(d * d) / (2.0f * var) - logf(o.sigma) - kLogSqrt2Pi;
  lp -= logf(1.0f + kTanhEps - o.t * o.t);
  lp -= logf(s);
  o.lp = lp;
  return o;
}

// backward of the above: given dL/da (gA) and dL/dlogp (gLp) returns dL/dmu, dL/draw.
DSACT_HD void tanh_gauss_bwd(float mu, float raw, float eps, float s, float lo_ls, float hi_ls, float gA,
                             float gLp, float& dmu, float& draw) {
  const float sigma = expf(clampf(raw, lo_ls, hi_ls));
  const float x = mu + eps * sigma;
  const float t = tanhf(x);
  const float omt2 = 1.0f - t * t;
  const float g = 2.0f * t * omt2 / (1.0f + kTanhEps - t * t);  // d logp / d x (tanh correction)
  const float dadx = s * omt2;
  dmu = gA * dadx + gLp * g;
  const float dsigma = gA * dadx * eps + gLp * (-1.0f / sigma + eps * g);
  const bool inside = (raw >= lo_ls) && (raw <= hi_ls);  // clamp passes gradient on [lo, hi]
  draw = inside ? dsigma * sigma : 0.0f;
}

DSACT_HD float huber50(float d) {
  const float ad = fabsf(d);
  return ad <= 50.0f ? 0.5f * d * d : 50.0f * (ad - 25.0f);
}

// ---- critic, one sample, one of the twin heads (dsac_v2.py:255-302) ----------------------------
struct CriticTerm {
  float loss;   // ratio*(huber(q-tq) + std*(sd^2 - huber(q-tqb))/(sd+0.1)), before the batch mean
  float dq;     // d loss / d q     (before 1/B)
  float dstd;   // d loss / d std   (before 1/B)
};
DSACT_HD CriticTerm critic_term(float q, float stdv, float ms, float tq, float tqs) {
  CriticTerm o;
  const float sd = fmaxf(stdv, 0.0f);
  const float ratio = clampf((ms * ms) / (sd * sd + 0.1f), 0.1f, 10.0f);
  const float bound = 3.0f * ms;
  const float tqb = q + clampf(tqs - q, -bound, bound);
  const float d1 = q - tq;
  const float h2 = huber50(q - tqb);
  const float w = (sd * sd - h2) / (sd + 0.1f);
  o.loss = ratio * (huber50(d1) + stdv * w);
  o.dq = ratio * clampf(d1, -50.0f, 50.0f);
  o.dstd = ratio * w;
  return o;
}

// ---- Adam, torch single-tensor path ---------------------------------------------------------------
// step_size = lr/(1-b1^t), bc2_sqrt = sqrt(1-b2^t) are computed in double on the caller side exactly
// like torch does in Python floats.
DSACT_HD void adam_update(float& p, float& m, float& v, float g, float b1w /*1-beta1*/, float beta2,
                          float b2w /*1-beta2*/, float step_size, float bc2_sqrt, float eps) {
  m = m + b1w * (g - m);       // exp_avg.lerp_(grad, 1-beta1)
  v = v * beta2;               // exp_avg_sq.mul_(beta2)
  v = v + (b2w * g) * g;       //            .addcmul_(grad, grad, value=1-beta2): (value*t1)*t2
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p + ((-step_size) * m) / denom;  // param.addcdiv_(exp_avg, denom, value=-step_size): (value*t1)/t2
}

// p_targ.mul_(polyak); p_targ.add_((1-polyak)*p)  -- three separately rounded fp32 ops
DSACT_HD float polyak_update(float pt, float p, float polyak, float one_minus) {
  const float a = pt * polyak;
  const float b = one_minus * p;
  return a + b;
}

}  // namespace dsact
