"""KD-tree build on the device (CorrespondenceFinderKDTree2D::reset, LSM2D_FINDER_KDTREE): wall time of the first finder call on a fresh
cloud set (build + one query call) against the second (query call alone), for a map-sized cloud and for a batch of scans, with both
forms of the sequential sums ("kd_chain" 1 systolic DPP pass, 0 one v_readlane per value).  One JSON line per case.
    python tools/kd_build_bench.py [--map-points N] [--scans S]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from srrg2_laser_slam_2d_amd import api, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--map-points", type=int, nargs="*", default=[10000, 100000, 1000000])
    ap.add_argument("--scans", type=int, default=1000)
    args = ap.parse_args()
    ctx = api.Context(0)
    wl = synth.make_workload(args.scans, 1000, seed=0)
    q = wl.scan_points[: wl.scan_offsets[1]]
    for chain in (1, 0):
        ctx.set_option("kd_chain", chain)
        for n in args.map_points:
            m = synth.make_map(synth.make_world(0), n, seed=0)
            for rep in range(2):
                cs = api.CloudSet(ctx, m)
                f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5, search="kdtree")
                f.setFixed(cs); f.setMoving(q); f.setLocalMapInSensor(synth.invert_poses(wl.x0[:1].astype(np.float64))[0].astype(np.float32))
                ctx.synchronize(); t0 = time.perf_counter(); f.compute(); t1 = time.perf_counter(); f.compute(); t2 = time.perf_counter()
                out = {"case": "map", "points": n, "kd_chain": chain, "rep": rep, "build_plus_query_ms": (t1 - t0) * 1e3, "query_ms": (t2 - t1) * 1e3,
                       "levels": ctx.get_option("last_kd_levels"), "nodes": ctx.get_option("last_kd_nodes")}
                cs.close()
            print(json.dumps(out), flush=True)
        for rep in range(2):
            cs = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
            f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5, search="kdtree")
            f.setFixed(cs, 0); f.setMoving(q); f.setLocalMapInSensor([0, 0, 0])
            ctx.synchronize(); t0 = time.perf_counter(); f.compute(); t1 = time.perf_counter(); f.compute(); t2 = time.perf_counter()
            out = {"case": "scans", "clouds": args.scans, "points": int(wl.scan_offsets[-1]), "kd_chain": chain, "rep": rep,
                   "build_plus_query_ms": (t1 - t0) * 1e3, "query_ms": (t2 - t1) * 1e3, "levels": ctx.get_option("last_kd_levels"), "nodes": ctx.get_option("last_kd_nodes")}
            cs.close()
        print(json.dumps(out), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
