"""How long are the culled stream's unit lists, and how often are they rebuilt?  (GPU box, DIAGNOSTICS BUILD: the library must be built with
LSM2D_EXTRA_HIPCC_FLAGS=-DLSM2D_DEBUG_UNITS, which puts the list length / rebuild flag of slice 0 in place of the outlier statistics.)
usage: LSM2D_EXTRA_HIPCC_FLAGS=-DLSM2D_DEBUG_UNITS python -m srrg2_laser_slam_2d_amd.build --force && python tools/units_probe.py"""
import numpy as np, math, sys
sys.path.insert(0, '.')
from srrg2_laser_slam_2d_amd import api, synth
wl = synth.make_workload(64, 100000, seed=0)
ctx = api.Context(0)
ctx.set_option("align_path", 1)
proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
fx = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mv = api.CloudSet(ctx, wl.map_points)
r = al.compute_batch([fx], [mv], wl.x0, want_stats=True)
u = r.stats["n_outliers"].astype(float); rb = r.stats["chi_outliers"]
print("units per iteration (mean over 64 alignments):", np.round(u.mean(0)).astype(int).tolist())
print("fraction of 3584:", np.round(u.mean(0) / 3584, 3).tolist())
print("rebuild flag at end of iteration (mean):", np.round(rb.mean(0), 2).tolist())
print("overall mean fraction", u.mean() / 3584)
