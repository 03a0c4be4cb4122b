import math, time, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
from srrg2_laser_slam_2d_amd import api, synth
ctx = api.Context(0)
wl = synth.make_workload(4, 100000, seed=0)
scan = api.CloudSet(ctx, wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]].copy()); mp = api.CloudSet(ctx, wl.map_points)
x0 = wl.x0[1]; inv = synth.invert_poses(x0[None, :].astype(np.float64))[0].astype(np.float32)
for name, f in (("projective", api.CorrespondenceFinderProjective2f(ctx, api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0))),
                ("nn", api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3)), ("distmap", api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05))):
    for role in ("A", "B"):
        if name == "projective" and role == "B": continue
        f.setFixed(scan if role == "A" else mp); f.setMoving(mp if role == "A" else scan); f.setLocalMapInSensor(x0 if role == "A" else inv)
        f.compute(); t = []
        for _ in range(50):
            t0 = time.perf_counter(); p = f.compute(); t.append(time.perf_counter() - t0)
        print("%-10s role %s: %7.1f us per compute() (median of 50), %d pairs" % (name, role, np.median(t) * 1e6, len(p)))
