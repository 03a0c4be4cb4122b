#!/usr/bin/env python3
"""Latency of a handful of alignments against a big map: one workgroup per alignment (k_align) vs the split path.
    python tools/small_batch_bench.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import api, synth

ctx = api.Context(0)
proj = api.PointNormal2fProjectorPolar(1081, -np.pi, np.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj), min_num_correspondences=10))
rows = []
for n_map in (10000, 100000, 1000000):
    for n in (1, 4, 16, 64):
        wl = synth.make_workload(n, n_map, seed=1)
        fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
        row = {"map_points": n_map, "alignments": n}
        ref = None
        for name, path in (("fused", 1), ("split", 2), ("auto", 0)):
            ctx.set_option("align_path", path)
            for _ in range(3):
                r = al.compute_batch([fixed], [moving], wl.x0)
            t = time.perf_counter(); k = 0.0
            for _ in range(10):
                r = al.compute_batch([fixed], [moving], wl.x0); k += r.kernel_ms
            row[name + "_ms_wall"] = (time.perf_counter() - t) * 100.0; row[name + "_ms_device"] = k / 10
            ref = r.pose if ref is None else ref
            assert np.array_equal(ref, r.pose)
        ctx.set_option("align_path", 0)
        row["max_err_vs_truth_m"] = float(np.abs(r.pose - wl.x_true)[:, :2].max())
        rows.append(row); print(json.dumps(row), flush=True)
