#!/usr/bin/env python3
"""In-kernel clock and launch time of the headline step over the first seconds of a process (DVFS behaviour of the box):
one line per 25 steps.  usage: python tools/clock_trace.py [steps]"""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srrg2_laser_slam_2d_amd import api, synth


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
    ctx = api.Context(0, kernel_timing=True)
    wl = synth.make_workload(1000, 100000, seed=0)
    proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
    al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
    scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mp = api.CloudSet(ctx, wl.map_points)
    t0 = time.perf_counter(); rows = []
    for i in range(steps):
        r = al.compute_batch([scans], [mp], wl.x0)
        rows.append((time.perf_counter() - t0, r.kernel_ms, r.kernel_clock_mhz))
        if i == steps // 2:
            time.sleep(1.0)          # an idle second in the middle: does the clock fall back?
    a = np.array(rows)
    for k in range(0, steps, 25):
        b = a[k:k + 25]
        print("steps %4d-%4d  t=%.3f s  kernel %.3f ms  clock %.0f MHz  cycles/launch %.3f M" % (k, k + len(b) - 1, b[0, 0], b[:, 1].mean(), b[:, 2].mean(), (b[:, 1] * b[:, 2]).mean() * 1e-3))


if __name__ == "__main__":
    main()
