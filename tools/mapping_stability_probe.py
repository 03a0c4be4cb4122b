"""Is the dispatcher's workgroup -> CU placement of a k_align launch the same from launch to launch?  (GPU box; what a placement by measured
mapping would have to rely on.)  usage: python tools/mapping_stability_probe.py"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dump = tempfile.mktemp(suffix=".stamps"); os.environ["LSM2D_DUMP_STAMPS"] = dump
from srrg2_laser_slam_2d_amd import api, synth
wl = synth.make_workload(1000, 100000, seed=0)
ctx = api.Context(0); ctx.set_option("clock_stride", 1); ctx.set_option("kernel_timing", 1); ctx.set_option("balance", 0)
proj = api.PointNormal2fProjectorPolar(1081, -np.pi, np.pi, 0.3, 30.0)
al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj), min_num_correspondences=10))
fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
for _ in range(6):
    al.compute_batch([fixed], [moving], wl.x0)
launches = open(dump).read().split("# launch")[1:]
maps = []
for ln in launches:
    rows = np.array([[int(x, 0) for x in r.split()] for r in ln.strip().splitlines()[1:]], dtype=np.int64)
    wg, cyc, ticks, start, hw = rows.T
    xcc = hw >> 32; hwid = hw & 0xffffffff
    cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
    m = np.zeros(1000, np.int64); m[wg] = cu; maps.append(m)
for i in range(1, len(maps)):
    same = (maps[i] == maps[i - 1]).mean()
    # same PARTITION (which workgroups share a CU), whatever the CU's name
    def part(m):
        d = {}
        for w, c in enumerate(m.tolist()): d.setdefault(c, []).append(w)
        return {tuple(v) for v in d.values()}
    print("launch %d vs %d: same CU for %.1f %% of the workgroups; identical groups %d of %d" % (i, i - 1, 100 * same, len(part(maps[i]) & part(maps[i - 1])), len(part(maps[i]))))
os.unlink(dump)
