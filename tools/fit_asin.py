"""Fit the fixed polynomial of the projector's octant angle (oracle and HIP kernels evaluate the SAME float32 operation sequence).

The column of a point needs phi = angle of (mx, mn) = (max, min) of (|x|, |y|), phi in [0, pi/4].  Round 1 took phi = atan(mn / mx):
one v_rcp_f32 for the divide and one v_rsq_f32 for the depth r = sqrt(x^2 + y^2) per point -- and a transcendental costs ~18 cycles
of a saturated SIMD in this stream (tools/valu_issue_probe.hip).  With t = mn / r = sin(phi), t in [0, sqrt(1/2)], the divide can
start from the depth's own v_rsq_f32 seed: ONE transcendental per point.
asin(t) = t + t*s*P(s), s = t*t in [0, 1/2]; the singularity (t = 1, s = 1) sits as far from [0, 1/2] as atan's (a = i) does from
[0, 1], so the same degree gives the same accuracy.
Run:  python tools/fit_asin.py
"""
import numpy as np

SMAX = 0.5 * (1 + 2e-7)      # t = fl(mn / r) may exceed sqrt(1/2) by an ulp


def fit(deg):
    n = 4000
    k = np.arange(n)
    s = SMAX * 0.5 * (1 - np.cos(np.pi * (k + 0.5) / n))
    t = np.sqrt(s)
    f = (np.arcsin(t) / t - 1.0) / s
    w = t * s
    V = np.vander(s, deg + 1, increasing=True)
    lw = np.ones_like(s)
    for _ in range(80):
        coef, *_ = np.linalg.lstsq(V * (w * lw)[:, None], f * w * lw, rcond=None)
        err = np.abs((V @ coef - f) * w)
        lw = lw * (err / err.max() + 1e-3) ** 0.5
        lw /= lw.max()
    return coef


def f32_eval(t32, coef32):
    t = t32.astype(np.float32)
    s = (t * t).astype(np.float32)
    p = np.full_like(s, coef32[-1])
    for c in coef32[-2::-1]:
        p = (p.astype(np.float64) * s.astype(np.float64) + np.float64(c)).astype(np.float32)
    ts = (t * s).astype(np.float32)
    return (ts.astype(np.float64) * p.astype(np.float64) + t.astype(np.float64)).astype(np.float32)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    tmax = np.sqrt(SMAX)
    t = np.concatenate([rng.random(4_000_000) * tmax, np.linspace(0, tmax, 2_000_001)]).astype(np.float32)
    for deg in (6, 7, 8):
        c = fit(deg)
        c32 = c.astype(np.float32)
        r = f32_eval(t, c32)
        err = np.abs(r.astype(np.float64) - np.arcsin(t.astype(np.float64)))
        print(deg, "max abs err %.3e rad" % err.max())
        print("   ", ", ".join("%.9ef" % v for v in c32))
