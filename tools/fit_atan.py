"""Fit the fixed polynomial used by lsm2d_atan2f (oracle and HIP kernels evaluate the SAME
float32 operation sequence, so column indices agree bit-for-bit between CPU and GPU).

atan(a), a in [0,1]:  atan(a) = a + a*s*P(s),  s = a*a,  P of degree DEG (Horner, fmaf).
Coefficients: Chebyshev-node least squares in float64 then rounded to float32; the script
prints the max abs error of the float32 evaluation against float64 atan.
Run:  python tools/fit_atan.py
"""
import numpy as np

def fit(deg):
    n = 4000
    k = np.arange(n)
    s = 0.5 * (1 - np.cos(np.pi * (k + 0.5) / n))          # Chebyshev nodes on [0,1]
    a = np.sqrt(s)
    f = (np.arctan(a) / a - 1.0) / s
    # minimise abs error of a*s*P(s): weight by a*s
    w = a * s
    V = np.vander(s, deg + 1, increasing=True)
    coef, *_ = np.linalg.lstsq(V * w[:, None], f * w, rcond=None)
    # a few Remez-like reweighting rounds (Lawson) to flatten the error
    lw = np.ones_like(s)
    for _ in range(60):
        coef, *_ = np.linalg.lstsq(V * (w * lw)[:, None], f * w * lw, rcond=None)
        err = np.abs((V @ coef - f) * w)
        lw = lw * (err / err.max() + 1e-3) ** 0.5
        lw /= lw.max()
    return coef

def f32_eval(a32, coef32):
    """float32 Horner with fma emulated through float64 (exact product, one rounding)."""
    a = a32.astype(np.float32)
    s = (a * a).astype(np.float32)
    p = np.full_like(s, coef32[-1])
    for c in coef32[-2::-1]:
        p = (p.astype(np.float64) * s.astype(np.float64) + np.float64(c)).astype(np.float32)
    t = (a * s).astype(np.float32)
    return (t.astype(np.float64) * p.astype(np.float64) + a.astype(np.float64)).astype(np.float32)

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.random(4_000_000), np.linspace(0, 1, 2_000_001)]).astype(np.float32)
    for deg in (6, 7, 8, 9):
        c = fit(deg)
        c32 = c.astype(np.float32)
        r = f32_eval(a, c32)
        err = np.abs(r.astype(np.float64) - np.arctan(a.astype(np.float64)))
        print(deg, "max abs err %.3e rad" % err.max(), "(double-poly err %.3e)" % np.abs(
            a.astype(np.float64) + a.astype(np.float64) ** 3 * np.polyval(c[::-1], a.astype(np.float64) ** 2)
            - np.arctan(a.astype(np.float64))).max())
        print("   ", ", ".join("%.9ef" % v for v in c32))
        print("   hex:", ", ".join(float(v).hex() for v in c32))
