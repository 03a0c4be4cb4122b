#!/usr/bin/env python3
"""Cycles per record of the reference-order walker ("sum_order" 1), in isolation: lsm2d_linearize over a long correspondence vector runs ONE workgroup
(k_linearize_seq), whose time is the walk (the 512 producing threads' work per trip is parallel and small).  python tools/seq_walk_probe.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import api, synth

wl = synth.make_workload(1, 100000, seed=0)
ctx = api.Context(0)
ctx.set_option("sum_order", 1)
f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
rng = np.random.default_rng(1)
for n in (4096, 65536, 262144):
    corr = np.stack([rng.integers(0, len(f), n), rng.integers(0, len(wl.map_points), n)], 1).astype(np.int32)
    fx = api.CloudSet(ctx, f); mv = api.CloudSet(ctx, wl.map_points)
    ms = []
    for _ in range(5):
        api.linearize(ctx, api.make_slice_params(), fx, mv, corr, wl.x0[0]); ms.append(ctx.last_kernel_ms())
    t = float(np.median(ms))
    print("n %7d pairs: %.3f ms, %.1f ns per record (~%.0f cycles at 2.2 GHz)" % (n, t, t * 1e6 / n, t * 1e6 / n * 2.2))
