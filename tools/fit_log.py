"""Fit the fixed polynomial of lsm2d's logf (the Cauchy kernel's statistic chi_out = tau * log(1 + chi/tau)): oracle and HIP kernels
evaluate the SAME float32 operation sequence, so the last libm call on the path goes away and the statistic is bit-identical too.

x = m * 2^e with m in [sqrt(1/2), sqrt(2)), f = m - 1:  log(x) = e*ln2 + f - f^2/2 + f^3 * P(f),  P of degree DEG (Horner, fmaf).
Coefficients: Chebyshev-node least squares in float64 with Lawson reweighting, rounded to float32.
Run:  python tools/fit_log.py
"""
import numpy as np

DEG = 7
LN2 = np.float32(0.6931471805599453)


def fit(deg=DEG):
    n = 6000
    k = np.arange(n)
    lo, hi = np.sqrt(0.5) - 1.0, np.sqrt(2.0) - 1.0
    f = 0.5 * (lo + hi) + 0.5 * (hi - lo) * np.cos(np.pi * (k + 0.5) / n)
    f = f[np.abs(f) > 1e-4]
    g = (np.log1p(f) - f + 0.5 * f * f) / f ** 3
    w = np.abs(f) ** 3
    V = np.vander(f, deg + 1, increasing=True)
    lw = np.ones_like(f)
    for _ in range(80):
        coef, *_ = np.linalg.lstsq(V * (w * lw)[:, None], g * w * lw, rcond=None)
        err = np.abs((V @ coef - g) * w)
        lw = lw * (err / err.max() + 1e-3) ** 0.5
        lw /= lw.max()
    return coef


def fma32(a, b, c):
    return (np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def log32(x32, c32):
    x = np.asarray(x32, np.float32)
    bits = x.view(np.uint32).astype(np.int64)
    e = (bits >> 23) - 127
    m = ((bits & 0x7FFFFF) | 0x3F800000).astype(np.uint32).view(np.float32)
    big = m > np.float32(1.41421354)
    m = np.where(big, (m * np.float32(0.5)).astype(np.float32), m); e = np.where(big, e + 1, e)
    f = (m - np.float32(1.0)).astype(np.float32)
    z = (f * f).astype(np.float32)
    p = np.full_like(f, c32[-1])
    for c in c32[-2::-1]:
        p = fma32(p, f, np.full_like(f, c))
    r = fma32((z * f).astype(np.float32), p, fma32(np.full_like(f, np.float32(-0.5)), z, f))
    return fma32(e.astype(np.float32), np.full_like(f, LN2), r)


if __name__ == "__main__":
    c = fit(); c32 = c.astype(np.float32)
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(1.0, 4.0, 2_000_000), np.exp(rng.uniform(0.0, 80.0, 2_000_000)), np.linspace(1.0, 2.0, 1_000_001)]).astype(np.float32)
    got = log32(x, c32).astype(np.float64); ref = np.log(x.astype(np.float64))
    print("x in [1, 5e34]: max abs err %.3e, max rel err (log >= 0.1) %.3e" % (np.abs(got - ref).max(), (np.abs(got - ref) / np.maximum(ref, 0.1)).max()))
    print("P:", ", ".join("%.10ef" % v for v in c32))
