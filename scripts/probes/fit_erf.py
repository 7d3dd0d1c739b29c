#!/usr/bin/env python3
"""Regenerates the erf coefficients of dsac-v2_amd/csrc/dsact_math.h (kErfS / kErfL) and their error figures.
CPU only (numpy + scipy). Chebyshev-node interpolation in double on each interval, rounded to fp32, then the fp32
Horner/FMA evaluation is emulated and compared with scipy's erf over 3M points."""
import numpy as np
from numpy.polynomial import chebyshev as Ch, polynomial as P
from scipy.special import erf, erfc

T0, THI = 0.9375, 4.1
f32 = np.float32


def fit(f, lo, hi, deg):
    k = np.arange(deg + 1)
    x = np.cos((2 * k + 1) * np.pi / (2 * (deg + 1)))
    c = Ch.chebfit(x, f(0.5 * (hi - lo) * x + 0.5 * (hi + lo)), deg)
    p = Ch.cheb2poly(c)
    a, b = 2 / (hi - lo), -(hi + lo) / (hi - lo)
    q, pw = np.zeros(1), np.ones(1)
    for ci in p:
        q = P.polyadd(q, ci * pw)
        pw = P.polymul(pw, np.array([b, a]))
    return q


def s_target(s):
    a = np.sqrt(np.maximum(s, 1e-300))
    return np.where(s < 1e-12, 2 / np.sqrt(np.pi) - 1, erf(a) / a - 1)


def l_target(t):
    return (np.log(erfc(t)) + t) / t


def horner(q, x):
    r = np.full_like(x, q[-1])
    for c in q[-2::-1]:
        r = (r.astype(np.float64) * x + np.float64(c)).astype(f32)  # one rounding per step, like fmaf
    return r


def erf32(a, qs, ql):
    a = a.astype(f32)
    t = np.minimum(np.abs(a), f32(THI))
    small = (horner(qs, a * a).astype(np.float64) * a + a).astype(f32)
    arg = (horner(ql, t).astype(np.float64) * t - t).astype(f32)
    e = np.exp2((arg * f32(1.4426950408889634)).astype(f32)).astype(f32)
    return np.where(t > f32(T0), np.copysign(f32(1) - e, a), small)


if __name__ == "__main__":
    qs = fit(s_target, 0.0, T0 * T0, 5).astype(f32)
    ql = fit(l_target, T0, THI, 8).astype(f32)
    a = np.concatenate([np.linspace(-6, 6, 2000001), np.random.default_rng(0).standard_normal(1000000) * 1.5]).astype(f32)
    ref = erf(a.astype(np.float64))
    err = np.abs(erf32(a, qs, ql).astype(np.float64) - ref)
    ulp = np.spacing(np.abs(ref.astype(f32))).astype(np.float64)
    print("kErfS:", ", ".join("%.9ef" % c for c in qs))
    print("kErfL:", ", ".join("%.9ef" % c for c in ql))
    print("max |err| %.3e at x=%g; max ulp error %.2f" % (err.max(), a[err.argmax()], (err / ulp).max()))
