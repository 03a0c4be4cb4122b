"""What decides when a CU is done with its four alignments?  (GPU box, DIAGNOSTICS BUILD -DLSM2D_DEBUG_UNITS: the statistics then carry the length of
slice 0's unit list per iteration.)  configs[1] with the placement OFF (alignment b runs in workgroup b), every workgroup stamped (start, lifetime,
CU): the four workgroups of a CU share its SIMDs and end together, so a CU's end time against the SUM over its alignments of a candidate cost model
says how good that model is as the thing to balance.
usage: LSM2D_EXTRA_HIPCC_FLAGS=-DLSM2D_DEBUG_UNITS python -m srrg2_laser_slam_2d_amd.build --force && python tools/balance_probe.py"""
import math, os, sys, tempfile
import numpy as np
sys.path.insert(0, '.')
dump = tempfile.mktemp(suffix=".stamps"); os.environ["LSM2D_DUMP_STAMPS"] = dump
from srrg2_laser_slam_2d_amd import api, synth
wl = synth.make_workload(1000, 100000, seed=0)
ctx = api.Context(0)
ctx.set_option("clock_stride", 1); ctx.set_option("balance", 0)
proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
fx = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mv = api.CloudSet(ctx, wl.map_points)
est = np.asarray(al.estimate_work([fx], [mv], wl.x0), float)
for _ in range(40):
    r = al.compute_batch([fx], [mv], wl.x0, want_stats=True)
launches = open(dump).read().split("# launch")[1:]
rows = np.array([[int(x, 0) for x in ln.split()] for ln in launches[-1].strip().splitlines()[1:]], dtype=np.int64)
wg, cyc, ticks, start, hw = rows.T
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
end = (start + ticks) * 0.01      # us
u = r.stats["n_outliers"].astype(float)            # units per iteration
rb = r.stats["chi_outliers"].astype(float)          # rebuild flag raised at the end of the iteration
units_sum = u.sum(1); rebuilds = 1 + rb[:, :-1].sum(1); ncorr = r.stats["n_correspondences"].astype(float).sum(1)
cus = np.unique(cu)
T = np.array([end[cu == c].max() for c in cus]); n_on = np.array([(cu == c).sum() for c in cus])
def per_cu(v):
    return np.array([v[wg[cu == c]].sum() for c in cus])
print("launch: %.1f us; CU end times: mean %.1f, min %.1f, max %.1f; workgroups per CU %s" % (end.max(), T.mean(), T.min(), T.max(), np.bincount(n_on).tolist()))
models = {"estimate (chunks at the start pose, margins)": per_cu(est), "units streamed, all iterations": per_cu(units_sum),
          "units + rebuilds": None, "units + rebuilds + pairs + per-workgroup constant": None}
X1 = np.stack([per_cu(units_sum), np.ones(len(cus))], 1)
X2 = np.stack([per_cu(units_sum), per_cu(rebuilds), np.ones(len(cus))], 1)
X3 = np.stack([per_cu(units_sum), per_cu(rebuilds), per_cu(ncorr), n_on.astype(float), np.ones(len(cus))], 1)
X0 = np.stack([per_cu(est), np.ones(len(cus))], 1)
for name, X in (("estimate (chunks at the start pose, with margins) + const", X0), ("units streamed over all iterations + const", X1), ("+ rebuilds", X2), ("+ pairs + workgroups on the CU", X3)):
    coef, res, *_ = np.linalg.lstsq(X, T, rcond=None)
    pred = X @ coef
    print("%-62s residual rms %.2f %% of the mean, max %.2f %%; coefficients %s" % (name, 100 * np.sqrt(np.mean((T - pred) ** 2)) / T.mean(), 100 * np.abs(T - pred).max() / T.mean(), np.round(coef, 4).tolist()))
print("spread of the CU sums themselves (no placement): units rms %.2f %% of their mean" % (100 * per_cu(units_sum).std() / per_cu(units_sum).mean()))
# the same launch WITH the placement: how equal are the sums it makes, and the end times it gets?
ctx.set_option("balance", 1)
open(dump, "w").close()
for _ in range(10):
    r = al.compute_batch([fx], [mv], wl.x0, want_stats=True)
launches = open(dump).read().split("# launch")[1:]
rows = np.array([[int(x, 0) for x in ln.split()] for ln in launches[-1].strip().splitlines()[1:]], dtype=np.int64)
slot, cyc, ticks, start, hw = rows.T      # (with a placement the stamp's row is the ALIGNMENT: k_align stamps a = order[workgroup])
xcc = hw >> 32; hwid = hw & 0xffffffff
cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
end = (start + ticks) * 0.01
cus = np.unique(cu)
T = np.array([end[cu == c].max() for c in cus]); n_on = np.array([(cu == c).sum() for c in cus])
S_est = np.array([est[slot[cu == c]].sum() for c in cus]); S_units = np.array([units_sum[slot[cu == c]].sum() for c in cus])
print("with the placement: launch %.1f us; CU end times mean %.1f min %.1f max %.1f (max / mean %.3f)" % (end.max(), T.mean(), T.min(), T.max(), T.max() / T.mean()))
for k in (3, 4):
    m = n_on == k
    if m.any():
        print("  CUs with %d workgroups: %d; sum of estimates mean %.1f (min %.1f max %.1f); sum of true units mean %.0f (rms %.2f %%); end time mean %.1f max %.1f" %
              (k, m.sum(), S_est[m].mean(), S_est[m].min(), S_est[m].max(), S_units[m].mean(), 100 * S_units[m].std() / S_units[m].mean(), T[m].mean(), T[m].max()))
print("  correlation of a CU's end time with its sum of true units: %.3f; with its sum of estimates: %.3f" % (np.corrcoef(T, S_units)[0, 1], np.corrcoef(T, S_est)[0, 1]))

# is what is left a property of the PLACE?  residual of "end = const + slope x units" per XCD and per shader engine
X = np.stack([S_units, np.ones(len(cus))], 1); coef, *_ = np.linalg.lstsq(X, T, rcond=None); resid = T - X @ coef
xcd_of = cus >> 16; se_of = (cus >> 8) & 7
print("  residual of the units model, balanced launch: rms %.2f %%, max %.2f %%" % (100 * resid.std() / T.mean(), 100 * np.abs(resid).max() / T.mean()))
print("  per XCD  : mean end time " + " ".join("%d:%.0f" % (x, T[xcd_of == x].mean()) for x in np.unique(xcd_of)) + " | mean residual " + " ".join("%+.1f" % resid[xcd_of == x].mean() for x in np.unique(xcd_of)))
print("  per SE   : mean residual " + " ".join("%d:%+.1f" % (x, resid[se_of == x].mean()) for x in np.unique(se_of)))
worst = np.argsort(-T)[:8]
print("  the eight CUs that end last: " + ", ".join("xcd %d se %d cu %d: %.0f us, units %+.1f %% of mean, %d wgs" % (xcd_of[i], se_of[i], cus[i] & 255, T[i], 100 * (S_units[i] / S_units[n_on == 4].mean() - 1), n_on[i]) for i in worst))
