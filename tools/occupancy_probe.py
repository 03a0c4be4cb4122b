"""How many k_align workgroups share a CU, and in how many dispatch rounds a batch runs (GPU box).

Every workgroup stamps its start tick, its lifetime and the hardware id of the CU it ran on (LSM2D_DUMP_STAMPS, clock_stride 1);
this script runs configs[1] once that way and prints: workgroups per CU, start-time histogram (dispatch rounds), lifetimes per round.
usage: python tools/occupancy_probe.py [--scans 1000] [--map-points 100000]
"""
import argparse
import collections
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=1000); ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20)
    a = ap.parse_args()
    dump = tempfile.mktemp(suffix=".stamps")
    os.environ["LSM2D_DUMP_STAMPS"] = dump
    from srrg2_laser_slam_2d_amd import api, synth
    wl = synth.make_workload(a.scans, a.map_points, seed=0)
    ctx = api.Context(0)
    ctx.set_option("clock_stride", 1)
    proj = api.PointNormal2fProjectorPolar(1081, -np.pi, np.pi, 0.3, 30.0)
    finder = api.CorrespondenceFinderProjective2f(ctx, proj)
    al = api.MultiAligner2D(ctx, max_iterations=a.iterations, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10))
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    for _ in range(3):
        res = al.compute_batch([fixed], [moving], wl.x0)
    print("kernel_ms %.3f clock %.0f MHz" % (res.kernel_ms, res.kernel_clock_mhz))
    launches = open(dump).read().split("# launch")[1:]
    rows = np.array([[int(x, 0) for x in ln.split()] for ln in launches[-1].strip().splitlines()[1:]], dtype=np.int64)
    wg, cyc, ticks, start, hw = rows.T
    xcc = hw >> 32; hwid = hw & 0xffffffff
    cu = (xcc << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
    # which workgroup ids share a CU: the dispatcher's dealing order (speed only -- the balanced ordering of lsm2d_align_batch assumes one)
    by_cu = collections.defaultdict(list)
    for w_, c_ in zip(wg.tolist(), cu.tolist()):
        by_cu[c_].append(w_)
    diffs = collections.Counter()
    for ids in by_cu.values():
        ids.sort()
        for a_, b_ in zip(ids[:-1], ids[1:]):
            diffs[b_ - a_] += 1
    print("workgroup-id distance between neighbours on one CU (most common):", diffs.most_common(6))
    print("first CUs:", [sorted(v) for v in list(by_cu.values())[:6]])
    print("XCD of workgroups 0..15:", xcc[np.argsort(wg)][:16].tolist())
    per_cu = collections.Counter(cu.tolist())
    print("CUs used %d; workgroups per CU over the launch: %s" % (len(per_cu), sorted(collections.Counter(per_cu.values()).items())))
    first = start < 0.2 * (start + ticks).max()
    per_cu_first = collections.Counter(cu[first].tolist())
    print("started in the first 20%% of the launch: %d workgroups; per CU: %s" % (int(first.sum()), sorted(collections.Counter(per_cu_first.values()).items())))
    end = (start + ticks).max()
    print("launch span %.3f ms (first start -> last end)" % (end * 1e-5))
    h, edges = np.histogram(start * 1e-5, bins=10, range=(0, end * 1e-5))
    print("start-time histogram [ms]:", ", ".join("%.2f:%d" % (e, c) for e, c in zip(edges[:-1], h)))
    for name, sel in (("first round", first), ("later", ~first)):
        if sel.any():
            print("%s: n=%d lifetime ms median %.3f min %.3f max %.3f; clock MHz median %.0f" % (
                name, int(sel.sum()), np.median(ticks[sel]) * 1e-5, ticks[sel].min() * 1e-5, ticks[sel].max() * 1e-5, np.median(cyc[sel] / ticks[sel]) * 100))
    ctx.close(); os.unlink(dump)


if __name__ == "__main__":
    main()
