import math, time, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from srrg2_laser_slam_2d_amd import api, synth
ctx = api.Context(0, kernel_timing=False)
wl = synth.make_workload(1000, 100000, seed=0)
proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mp = api.CloudSet(ctx, wl.map_points)
ref = None
for zc in (256, 4096, 256, 4096):
    ctx.set_option("zero_copy_max", zc)
    for _ in range(200): r = al.compute_batch([scans], [mp], wl.x0)
    t0 = time.perf_counter()
    for _ in range(300): r = al.compute_batch([scans], [mp], wl.x0)
    dt = (time.perf_counter() - t0) / 300
    if ref is None: ref = (r.pose.copy(), r.information.copy(), r.status.copy(), r.iterations.copy())
    same = all(np.array_equal(a, b) for a, b in zip(ref, (r.pose, r.information, r.status, r.iterations)))
    print("zero_copy_max %5d: %.4f ms per step = %.0f alignments/s, results identical: %s" % (zc, dt * 1e3, 1000 / dt, same))
