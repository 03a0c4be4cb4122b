"""How well does the placement's work estimate predict what an alignment streams?  (GPU box, DIAGNOSTICS BUILD -DLSM2D_DEBUG_UNITS: the statistics carry the length of
slice 0's unit list per iteration.)  configs[1]: per alignment the estimate (chunks kept at the start pose under the estimate's margins, lsm2d_estimate_work), the
block-level list at the start pose (iteration 0's units) and the units really streamed over the twenty iterations; residuals of linear fits of the last on each of the
first two -- what a block-level estimate would buy over the chunk-level one.
usage: LSM2D_EXTRA_HIPCC_FLAGS=-DLSM2D_DEBUG_UNITS python -m srrg2_laser_slam_2d_amd.build --force && python tools/estimate_quality_probe.py"""
import math, sys
import numpy as np
sys.path.insert(0, '.')
from srrg2_laser_slam_2d_amd import api, synth
wl = synth.make_workload(1000, 100000, seed=0)
ctx = api.Context(0)
ctx.set_option("align_path", 1)
proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
fx = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mv = api.CloudSet(ctx, wl.map_points)
est = np.asarray(al.estimate_work([fx], [mv], wl.x0), float)
r = al.compute_batch([fx], [mv], wl.x0, want_stats=True)
u = r.stats["n_outliers"].astype(float)
total = u.sum(1); u0 = u[:, 0]; u_last = u[:, -1]
# the same estimate made AT THE TRUE POSE (what a tracker with a good prior would have)
t_true = synth.invert_poses(wl.x_true).astype(np.float32) if hasattr(wl, "x_true") else None
def fit(x, y, name):
    A = np.stack([x, np.ones_like(x)], 1); c, *_ = np.linalg.lstsq(A, y, rcond=None); res = y - A @ c
    print("%-58s corr %.4f   residual rms %.2f %% of the mean   worst %+.1f %%" % (name, np.corrcoef(x, y)[0, 1], 100 * res.std() / y.mean(), 100 * np.abs(res).max() / y.mean()))
print("units streamed per alignment: mean %.0f, min %.0f, max %.0f (x %.2f)" % (total.mean(), total.min(), total.max(), total.max() / total.min()))
fit(est, total, "estimate (chunks at the start pose, estimate's margins)")
fit(u0, total, "block-level list at the start pose (iteration 0's units)")
fit(u_last, total, "block-level list at the converged pose (last iteration)")
fit(est, u_last, "estimate -> converged list")
fit(u0, u_last, "start-pose list -> converged list")
if ctx.get_option("experiments") == 1:      # other margins of the estimate (experiments build: cull_est_um / cull_est_urad)
    E = {}
    for um, urad in ((0, 20000), (0, 40000), (0, 60000), (0, 80000), (20000, 40000), (50000, 40000), (50000, 60000), (100000, 60000), (100000, 100000), (50000, 0), (200000, 0)):
        ctx.set_option("cull_est_um", um); ctx.set_option("cull_est_urad", urad)
        e2 = np.asarray(al.estimate_work([fx], [mv], wl.x0), float)
        fit(e2, total, "estimate with margins %d um / %d urad (mean %.0f chunks)" % (um, urad, e2.mean()))
        E[(um, urad)] = e2
    def fit2(cols, name):
        A = np.stack(list(cols) + [np.ones(len(total))], 1); c, *_ = np.linalg.lstsq(A, total, rcond=None); res = total - A @ c
        print("%-58s residual rms %.2f %% of the mean   worst %+.1f %%   weights %s" % (name, 100 * res.std() / total.mean(), 100 * np.abs(res).max() / total.mean(), np.round(c, 2).tolist()))
    n_fixed = np.diff(wl.scan_offsets).astype(float)
    fit2([E[(0, 40000)], E[(0, 20000)]], "two estimates: 0/40000 + 0/20000")
    fit2([E[(0, 40000)], E[(0, 80000)]], "two estimates: 0/40000 + 0/80000")
    fit2([E[(0, 40000)], E[(200000, 0)]], "two estimates: 0/40000 + 200000/0")
    fit2([E[(0, 40000)], u0], "estimate 0/40000 + block-level list at the start pose")
    fit2([E[(0, 40000)], n_fixed], "estimate 0/40000 + points of the scan")
    fit2([E[(0, 40000)], u0, n_fixed], "estimate 0/40000 + start-pose list + points of the scan")
    fit2([E[(0, 20000)], E[(0, 40000)], E[(0, 80000)], E[(200000, 0)], u0, n_fixed], "all of them")
