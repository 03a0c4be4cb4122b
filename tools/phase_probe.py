#!/usr/bin/env python3
"""Where a k_align workgroup's cycles go (library built with -DLSM2D_PHASE_PROBE): query / projection phase, barrier + reduction, solve + rest.
usage: LSM2D_EXTRA_HIPCC_FLAGS=-DLSM2D_PHASE_PROBE python -m srrg2_laser_slam_2d_amd.build --force && python tools/phase_probe.py <role> <finder>"""
import math, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dump = tempfile.mktemp(suffix=".stamps"); os.environ["LSM2D_DUMP_STAMPS"] = dump
from srrg2_laser_slam_2d_amd import api, synth

role, kind = (sys.argv + ["B", "distmap"])[1:3]
ctx = api.Context(0, kernel_timing=True); ctx.set_option("clock_stride", 1)
for kv in filter(None, os.environ.get("LSM2D_BENCH_OPTIONS", "").split(",")):
    ctx.set_option(kv.partition("=")[0].strip(), int(kv.partition("=")[2]))
wl = synth.make_workload(1000, 100000, seed=0)
if kind == "projective":
    f = api.CorrespondenceFinderProjective2f(ctx, api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0), 0.5, 0.8)
elif kind == "nn":
    f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5 if role == "B" else 0.3)
else:
    f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(f, min_num_correspondences=10))
scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); mp = api.CloudSet(ctx, wl.map_points)
x0 = wl.x0 if role == "A" else synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32)
for _ in range(5):
    r = al.compute_batch([scans], [mp], x0) if role == "A" else al.compute_batch([mp], [scans], x0)
rows = [l.split() for l in open(dump) if not l.startswith("#")][-1000:]
a = np.array([[int(v) for v in r[1:4]] + [int(r[4], 16)] for r in rows], dtype=np.float64)
# columns as dumped: lifetime cycles, query cycles, (reduce cycles - t0), solve cycles -- the dump subtracts the smallest third column
life, q, solve = a[:, 0], a[:, 1], a[:, 3]
red = life - q - solve          # (the dumped third column is relative to its minimum: take the remainder instead)
print("role %s / %s: kernel %.3f ms; per workgroup (median over %d): lifetime %.0f kcyc = query %.0f + barrier wait / reduction / prologue %.0f + solve and next-iteration set-up %.0f kcyc"
      % (role, kind, r.kernel_ms, len(a), np.median(life) / 1e3, np.median(q) / 1e3, np.median(red) / 1e3, np.median(solve) / 1e3))
os.remove(dump)
