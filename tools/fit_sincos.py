"""Fit the fixed polynomials of lsm2d's sincos (oracle, host code and HIP kernels evaluate the SAME float32 operation sequence, so
the rotation of a pose has the same bits on the CPU and on the GPU -- the two libms differ in the last place now and then).

Reduction: k = rint(x * 2/pi); r = fma(-k, PIO2_HI, x); r = fma(-k, PIO2_LO, r)      (|x| up to a few thousand radians)
sin(r) = r + (r*z) * S(z),  cos(r) = fma(z*z, C(z), fma(-0.5, z, 1)),  z = r*r,  r in [-pi/4, pi/4];  S, C of degree 2.
Coefficients: Chebyshev-node least squares in float64 with Lawson reweighting, rounded to float32; the script prints the
max abs error of the float32 evaluation (fma emulated through float64) against float64 sin / cos over many arguments.
Run:  python tools/fit_sincos.py
"""
import numpy as np

PIO2_HI = np.float32(1.5707963705062866)                       # fp32(pi/2)
PIO2_LO = np.float32(np.pi / 2 - float(PIO2_HI))               # the rest
TWO_OVER_PI = np.float32(2.0 / np.pi)


def lawson(V, f, w, rounds=80):
    lw = np.ones_like(f)
    for _ in range(rounds):
        coef, *_ = np.linalg.lstsq(V * (w * lw)[:, None], f * w * lw, rcond=None)
        err = np.abs((V @ coef - f) * w)
        lw = lw * (err / err.max() + 1e-3) ** 0.5
        lw /= lw.max()
    return coef


def fit(deg_s=2, deg_c=2):
    n = 6000
    k = np.arange(n)
    r = (np.pi / 4) * np.cos(np.pi * (k + 0.5) / n)
    r = r[np.abs(r) > 1e-4]
    z = r * r
    fs = (np.sin(r) - r) / (r * z); ws = np.abs(r * z)
    fc = (np.cos(r) - (1.0 - 0.5 * z)) / (z * z); wc = z * z
    S = lawson(np.vander(z, deg_s + 1, increasing=True), fs, ws)
    Cc = lawson(np.vander(z, deg_c + 1, increasing=True), fc, wc)
    return S, Cc


def f32(x):
    return np.asarray(x, np.float64).astype(np.float32)


def fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + np.asarray(c, np.float64)).astype(np.float32)


def sincos32(x32, S32, C32):
    x = x32.astype(np.float32)
    kf = np.rint((x * TWO_OVER_PI).astype(np.float32)).astype(np.float32)
    r = fma32(-kf, np.full_like(x, PIO2_HI), x)
    r = fma32(-kf, np.full_like(x, PIO2_LO), r)
    z = (r * r).astype(np.float32)
    ps = np.full_like(z, S32[-1])
    for c in S32[-2::-1]:
        ps = fma32(ps, z, np.full_like(z, c))
    s = fma32((r * z).astype(np.float32), ps, r)
    pc = np.full_like(z, C32[-1])
    for c in C32[-2::-1]:
        pc = fma32(pc, z, np.full_like(z, c))
    c_ = fma32((z * z).astype(np.float32), pc, fma32(np.full_like(z, np.float32(-0.5)), z, np.ones_like(z)))
    q = kf.astype(np.int64) & 3
    sin = np.where(q == 0, s, np.where(q == 1, c_, np.where(q == 2, -s, -c_)))
    cos = np.where(q == 0, c_, np.where(q == 1, -s, np.where(q == 2, -c_, s)))
    return sin, cos


if __name__ == "__main__":
    S, Cc = fit()
    S32, C32 = S.astype(np.float32), Cc.astype(np.float32)
    rng = np.random.default_rng(0)
    for span in (np.pi / 4, np.pi, 7.0, 100.0, 3000.0):
        x = np.concatenate([rng.uniform(-span, span, 3_000_000), np.linspace(-span, span, 1_000_001)]).astype(np.float32)
        s, c = sincos32(x, S32, C32)
        es = np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max()
        ec = np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64))).max()
        print("|x| <= %-8.4g max abs err sin %.3e cos %.3e" % (span, es, ec))
    print("PIO2_HI %.10ef (%s)  PIO2_LO %.10ef (%s)  2/pi %.10ef" % (PIO2_HI, float(PIO2_HI).hex(), PIO2_LO, float(PIO2_LO).hex(), TWO_OVER_PI))
    print("S:", ", ".join("%.10ef" % v for v in S32), " hex:", ", ".join(float(v).hex() for v in S32))
    print("C:", ", ".join("%.10ef" % v for v in C32), " hex:", ", ".join(float(v).hex() for v in C32))
