import math, time, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from srrg2_laser_slam_2d_amd import api, synth
ctx = api.Context(0, kernel_timing=bool(int(os.environ.get("KT", "0"))))
wl = synth.make_workload(1000, 100000, seed=0)
proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mp = api.CloudSet(ctx, wl.map_points)
import gc; gc.collect(); gc.freeze()
for _ in range(300): al.compute_batch([scans], [mp], wl.x0)
t = []
for _ in range(3000):
    t0 = time.perf_counter(); al.compute_batch([scans], [mp], wl.x0); t.append(time.perf_counter() - t0)
t = np.array(t) * 1e3
print("timing=%s spin=%s: mean %.4f median %.4f p99 %.4f p99.9 %.4f max %.4f ms; steps over 1.2x median: %d of %d" % (os.environ.get("KT"), os.environ.get("LSM2D_SYNC_SPIN"), t.mean(), np.median(t), np.percentile(t, 99), np.percentile(t, 99.9), t.max(), (t > 1.2 * np.median(t)).sum(), len(t)))
