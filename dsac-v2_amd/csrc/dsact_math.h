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

// ---- GELU (exact-erf form) and its derivative -------------------------------------------------------
// erf(x), |error| <= 7.7e-8 (1.3 ulp): two-interval fit made for this kernel (Chebyshev-node fits in double, rounded
// to fp32; scripts/probes/fit_erf.py regenerates the coefficients and the error figures):
//   |x| <= 0.9375 : erf = x + x*S(x^2)                         (degree 5)
//   |x| >  0.9375 : erf = sign(x) * (1 - exp(-t + t*L(t))),  t = min(|x|, 4.1)   (degree 8; erf(4.1) == 1 in fp32)
// Branch-free on purpose: a tile stage runs ONE wave per SIMD, so the epilogue is a serial VALU chain -- both
// intervals are evaluated with FMAs (packed v_pk_fma_f32 in the 4-wide device form, gelu4 in dsact_kernels.h) and
// selected, instead of the library erff/expf with their range branches. exp goes through the hardware 2^x.
constexpr float kErfT0 = 0.9375f, kErfHi = 4.1f, kLog2e = 1.4426950408889634f;
constexpr float kErfS[6] = {1.283791512e-01f, -3.761247694e-01f, 1.128162965e-01f, -2.675944008e-02f, 4.982214887e-03f,
                            -5.933305947e-04f};
constexpr float kErfL[9] = {-1.287695765e-01f, -6.346949339e-01f, -1.069155484e-01f, 2.425363660e-02f, -3.802132560e-03f,
                            3.235395125e-04f,  2.627512004e-06f,  -3.285806315e-06f, 2.156770051e-07f};

DSACT_HD float exp2_hw(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_exp2f(x);  // v_exp_f32: ~1 ulp, denormal results flush to 0
#else
  return exp2f(x);
#endif
}

DSACT_HD float erf_2fit(float x) {
  const float t = fminf(fabsf(x), kErfHi);
  const float s = x * x;
  float rs = kErfS[5];
  for (int k = 4; k >= 0; --k) rs = fmaf(rs, s, kErfS[k]);
  const float small = fmaf(rs, x, x);
  float rl = kErfL[8];
  for (int k = 7; k >= 0; --k) rl = fmaf(rl, t, kErfL[k]);
  const float e = exp2_hw(fmaf(rl, t, -t) * kLog2e);
  const float large = copysignf(1.0f - e, x);
  return t > kErfT0 ? large : small;
}

// gelu(z) and d gelu / dz share the erf; both are stored by the forward epilogue.
DSACT_HD void gelu_fwd_grad(float z, float& h, float& g) {
  const float cdf = fmaf(erf_2fit(z * kInvSqrt2), 0.5f, 0.5f);
  const float pdf = kInvSqrt2Pi * exp2_hw((-0.5f * z) * z * kLog2e);
  h = z * cdf;
  g = fmaf(z, pdf, cdf);
}

// ---- hidden activations of the MLPs (utils/common_utils.py:16-45 -> nn.ReLU / ELU / GELU / SELU / Sigmoid / Tanh with
// their default arguments). Every forward epilogue stores h = act(z) AND g = act'(z); the backward only ever multiplies
// by the stored g, so the activation exists in exactly one place per kernel family.
enum : int { ACT_GELU = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SELU = 3, ACT_SIGMOID = 4, ACT_TANH = 5 };
constexpr float kSeluAlpha = 1.6732632423543772848170429916717f, kSeluScale = 1.0507009873554804934193349852946f;
DSACT_HD void act_fwd_grad(int act, float z, float& h, float& g) {
  switch (act) {
    case ACT_RELU: h = z > 0.0f ? z : 0.0f; g = z > 0.0f ? 1.0f : 0.0f; break;
    case ACT_ELU: {      // alpha = 1: torch backward uses the result: grad * (h + alpha) below 0
      const float e = expm1f(z);
      h = z > 0.0f ? z : e; g = z > 0.0f ? 1.0f : e + 1.0f; break;
    }
    case ACT_SELU: {
      const float e = expm1f(z);
      h = kSeluScale * (z > 0.0f ? z : kSeluAlpha * e);
      g = z > 0.0f ? kSeluScale : kSeluScale * kSeluAlpha * (e + 1.0f); break;
    }
    case ACT_SIGMOID: { const float sg = 1.0f / (1.0f + expf(-z)); h = sg; g = sg * (1.0f - sg); break; }
    case ACT_TANH: { const float t = tanhf(z); h = t; g = 1.0f - t * t; break; }
    default: gelu_fwd_grad(z, h, g); break;
  }
}

// ---- OUTPUT activations (value_output_activation / policy_output_activation, networks/mlp.py:15-20: the last Linear is
// followed by `output_activation()`; every shipped example uses "linear"). Code: 0 = linear, else an ACT_* id (relu, elu,
// selu, sigmoid, tanh) or OUT_ACT_GELU. The heads keep the POST-activation outputs y, and the derivative is expressed through y
// so that nothing else has to be stored -- except for GELU (round 6), whose derivative is no function of its output: the
// head that evaluates it stores d y / d z beside y (tile-stage kernels: HeadsArgs::qdmean / pi_dact).
enum : int { OUT_ACT_GELU = 6 };
DSACT_HD float out_act_fwd(int act, float z) {
  if (act == 0) return z;
  float h, g;
  act_fwd_grad(act == OUT_ACT_GELU ? ACT_GELU : act, z, h, g);
  return h;
}
DSACT_HD float out_act_grad_y(int act, float y) {
  switch (act) {
    case ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case ACT_ELU: return y > 0.0f ? 1.0f : y + 1.0f;
    case ACT_SELU: return y > 0.0f ? kSeluScale : y + kSeluScale * kSeluAlpha;
    case ACT_SIGMOID: return y * (1.0f - y);
    case ACT_TANH: return 1.0f - y * y;
    default: return 1.0f;
  }
}

// ... where the pre-activation z is at hand (the kernel that evaluates the head)
DSACT_HD float out_act_grad(int act, float y, float z) {
  if (act != OUT_ACT_GELU) return out_act_grad_y(act, y);
  float h, g;
  gelu_fwd_grad(z, h, g);
  return g;
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
// s == 0 selects the reference's plain GaussDistribution (utils/act_distribution_cls.py:82-115; policy_act_distribution =
// "GaussDistribution"): no squashing -- a = x, logp = Normal(mu, sigma).log_prob(x). (A tanh-Gaussian's half range is > 0:
// dsact_set_action_limits rejects high <= low, and stores 0 in every dimension when the handle was created for the Gaussian.)
DSACT_HD TanhGaussFwd tanh_gauss_fwd(float mu, float raw, float eps, float s, float c, float lo_ls,
                                     float hi_ls) {
  // Every product below is rounded on its own, as the reference's separate tensor ops are (the library is built with
  // -ffp-contract=off; the pragma keeps that true for this function whatever the build says): the tanh correction
  // log(1 + 1e-6 - t^2) of a saturated action divides by ~1e-6 .. 1e-3, where one rounding more or less of t*t moves it
  // (and the gradient through it) by up to 1e-4 relative.
#pragma clang fp contract(off)
  TanhGaussFwd o;
  o.sigma = expf(clampf(raw, lo_ls, hi_ls));
  const float x = mu + eps * o.sigma;  // Normal.rsample: loc + eps*scale
  const float d = x - mu;
  const float var = o.sigma * o.sigma;
  // Normal.log_prob: -((v-loc)^2)/(2 var) - log(scale) - log(sqrt(2 pi))
  float lp = -(d * d) / (2.0f * var) - logf(o.sigma) - kLogSqrt2Pi;
  if (s == 0.0f) { o.t = x; o.a = x; o.lp = lp; return o; }   // GaussDistribution.rsample (:99-102)
  o.t = tanhf(x);
  o.a = s * o.t + c;
  const float t2 = o.t * o.t;          // torch.pow(tanh(action), 2)
  lp -= logf((1.0f + kTanhEps) - t2);
  lp -= logf(s);
  o.lp = lp;
  return o;
}

// backward of the above: given dL/da (gA) and dL/dlogp (gLp) returns dL/dmu, dL/draw.
DSACT_HD void tanh_gauss_bwd(float mu, float raw, float eps, float s, float lo_ls, float hi_ls, float gA,
                             float gLp, float& dmu, float& draw) {
#pragma clang fp contract(off)
  const float sigma = expf(clampf(raw, lo_ls, hi_ls));
  const float x = mu + eps * sigma;
  if (s == 0.0f) {   // GaussDistribution: a = x, logp = -eps^2/2 - log(sigma) - const through x = mu + eps sigma
    dmu = gA;
    const float dsg = gA * eps + gLp * (-1.0f / sigma);
    draw = ((raw >= lo_ls) && (raw <= hi_ls)) ? dsg * sigma : 0.0f;
    return;
  }
  const float t = tanhf(x);
  const float omt2 = fmaf(-t, t, 1.0f);   // tanh backward: 1 - y*y in ONE rounding, which is what ATen's vectorised kernel does
                                          // (until round 3 this was 1.0f - t*t, two roundings: 6e-5 relative on saturated rows)
  const float t2 = t * t;                 // pow(t, 2): rounded, then subtracted from the rounded 1 + 1e-6
  const float g = 2.0f * t * omt2 / ((1.0f + kTanhEps) - t2);  // d logp / d x (tanh correction)
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
