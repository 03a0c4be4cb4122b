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
    ap.add_argument("--wide-min-points", type=int, default=-1, help="kd_wide_min_points (levels of a map-sized cloud built by a workgroup per node); -1: the default")
    args = ap.parse_args()
    ctx = api.Context(0)
    if args.wide_min_points >= 0:
        ctx.set_option("kd_wide_min_points", args.wide_min_points)
    wl = synth.make_workload(args.scans, 1000, seed=0)
    q = wl.scan_points[: wl.scan_offsets[1]]
    # the live tracker's case: ONE scan's tree per call (a fresh cloud every time, as CorrespondenceFinderKDTree2D::reset() sees it), with the
    # single-launch workgroup build (default) and with the level-by-level build of round 3 ("kd_wg_max_points" 0)
    for wg in (16384, 0):
        ctx.set_option("kd_wg_max_points", wg)
        ts = []
        for rep in range(min(30, args.scans)):
            sc = wl.scan_points[wl.scan_offsets[rep]:wl.scan_offsets[rep + 1]]
            cs = api.CloudSet(ctx, sc)
            f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5, search="kdtree")
            f.setFixed(cs); f.setMoving(q); f.setLocalMapInSensor([0, 0, 0])
            ctx.synchronize(); t0 = time.perf_counter(); f.compute(); t1 = time.perf_counter(); f.compute(); t2 = time.perf_counter()
            ts.append(((t1 - t0) - (t2 - t1)) * 1e3)
            cs.close()
        print(json.dumps({"case": "one scan per call", "kd_wg_max_points": wg, "build_ms_median": float(np.median(ts[5:])), "build_ms_min": float(np.min(ts[5:])),
                          "levels": ctx.get_option("last_kd_levels"), "nodes": ctx.get_option("last_kd_nodes")}), flush=True)
        cs = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
        f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5, search="kdtree")
        f.setFixed(cs, 0); f.setMoving(q); f.setLocalMapInSensor([0, 0, 0])
        ctx.synchronize(); t0 = time.perf_counter(); f.compute(); t1 = time.perf_counter(); f.compute(); t2 = time.perf_counter()
        print(json.dumps({"case": "scans", "clouds": args.scans, "kd_wg_max_points": wg, "build_ms": ((t1 - t0) - (t2 - t1)) * 1e3,
                          "levels": ctx.get_option("last_kd_levels"), "nodes": ctx.get_option("last_kd_nodes")}), flush=True)
        cs.close()
    ctx.set_option("kd_wg_max_points", 16384)
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
