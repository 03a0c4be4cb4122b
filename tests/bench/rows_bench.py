#!/usr/bin/env python3
"""Throughput of the rows either side of the hot path (SURVEY.md 8 rows f1, f2, a4 / a5 `reset()`), each NEXT TO the CPU oracle
on a sample of the same inputs and gated on bit-equality with it:

  f2   RawDataPreprocessorProjective2D over a batch of scans (sensor_processing/raw_data_preprocessor_projective_2d.cpp:59-107)
  a4   CorrespondenceFinderKDTree2D "reset": the search structure over a fixed cloud (registration/correspondence_finder_kd_tree_2d.cpp:6-10)
  a5   CorrespondenceFinderNN2D "reset": the distance map over a fixed cloud (registration/correspondence_finder_nn_2d.cpp:20-61)
  f1   SceneClipperProjective2D / MergerProjective2D on a map-sized scene (mapping/scene_clipper_projective_2d.cpp:12-73, merger_projective_2d.cpp:19-116)

One JSON line per row: wall milliseconds per call (median over --reps, synchronised), units per second, algorithmic bytes per
second, the oracle's rate on one host core, parity.  Per-kernel durations: run under
`rocprofv3 --kernel-trace --stats -- python3 tests/bench/rows_bench.py` (tools/profile_round.sh does).
    python tests/bench/rows_bench.py [--scans 8192] [--map-points 100000] [--reps 10]
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def timed(fn, reps, sync):
    fn(); sync()                                    # warm-up: allocations, first-use code load
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); sync(); t.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(t)), float(np.min(t))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scans", type=int, default=8192)
    ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--cpu-sample", type=int, default=64)
    args = ap.parse_args()
    from oracle import pyoracle as po
    from srrg2_laser_slam_2d_amd import api, synth

    po.lib()                                        # build / load the oracle before anything is timed against it
    ctx = api.Context(0, kernel_timing=True)
    sync = ctx.synchronize
    world = synth.make_world(0)
    out = []

    def emit(row, **kw):
        kw = {"row": row, **kw}
        out.append(kw)
        print(json.dumps(kw), flush=True)

    # ---- f2: ranges -> clouds, a batch of scans in one launch ---------------------------------------------------------------------
    n_unique = min(args.scans, 512)
    poses = synth.sample_poses(world, n_unique, seed=4)
    a0, a1 = -0.75 * math.pi, 0.75 * math.pi
    uniq = synth.make_scan_ranges(world, poses, n_beams=args.beams, angle_min=a0, angle_max=a1, noise_sigma=0.005, seed=1)
    ranges = np.ascontiguousarray(uniq[np.arange(args.scans) % n_unique])
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=30.0, voxelize_resolution=0.02, normal_point_distance=0.3)
    pre.setRawData(ranges, a0, a1, 0.0, 60.0)
    kernel_ms = []

    def run_pre():
        cs = pre.compute(); kernel_ms.append(ctx.last_kernel_ms()); run_pre.last = cs
    wall, best = timed(run_pre, args.reps, sync)
    meas = run_pre.last
    pp = po.Preprocessor(args.beams, a0, a1, 0.3, 30.0, 0.3, 5, 0.02)
    t0 = time.perf_counter(); ok = True
    ns = min(args.cpu_sample, args.scans)
    for i in range(ns):
        want = po.preprocess_scan(pp, ranges[i])
        ok = ok and meas.counts[i] == len(want) and np.array_equal(meas.download(i), want)
    cpu_s = time.perf_counter() - t0                       # includes the downloads; the oracle dominates
    k_ms = float(np.median(kernel_ms[1:]))
    alg = 4.0 * args.beams * args.scans + 16.0 * float(meas.counts.sum())
    # the same call with the ranges in pinned host memory (no staging pass) and already on the device (nothing crosses the host link)
    import torch
    walls = {}
    for where, src in (("pinned", torch.from_numpy(ranges).pin_memory()), ("device", torch.from_numpy(ranges).to("cuda:0"))):
        pre.setRawData(src, a0, a1, 0.0, 60.0)
        walls[where], _ = timed(run_pre, args.reps, sync)
        ok = ok and np.array_equal(run_pre.last.counts, meas.counts) and np.array_equal(run_pre.last.download(7), meas.download(7))
    emit("f2 preprocess_scans", scans=args.scans, beams=args.beams, points_out=int(meas.counts.sum()), wall_ms=wall, kernel_ms=k_ms,
         wall_ms_pinned_ranges=walls["pinned"], wall_ms_device_ranges=walls["device"], scans_per_s_wall_device_ranges=args.scans / (walls["device"] * 1e-3),
         scans_per_s_kernel=args.scans / (k_ms * 1e-3), scans_per_s_wall_host_ranges_in=args.scans / (wall * 1e-3),
         algorithmic_GBs_kernel=alg / (k_ms * 1e-3) / 1e9, frac_of_hbm_peak=alg / (k_ms * 1e-3) / 8e12,
         cpu_port_scans_per_s_1core=ns / cpu_s, parity_bit_identical=bool(ok), sample=ns)

    # ---- a4 / a5 reset: search structures over the fixed clouds ---------------------------------------------------------------------
    wl = synth.make_workload(1000, args.map_points, seed=0)
    scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    mp = api.CloudSet(ctx, wl.map_points)
    q = api.CloudSet(ctx, wl.map_points[:64].copy())      # a token moving cloud: the call's cost is the structure it has to build first
    for name, finder, osp in (
            ("a4 grid over 1000 scans + 1 query call", api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3), dict(finder=po.FINDER_NN, max_distance=0.3)),
            ("a5 distance maps over 1000 scans + 1 query call", api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05), dict(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.05))):
        def build():
            fresh = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)      # structures are cached per set: a new set rebuilds them
            finder.setFixed(fresh, 3); finder.setMoving(q); finder.setLocalMapInSensor(wl.x0[3]); build.pairs = finder.compute(); fresh.close()

        def upload_only():
            fresh = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); fresh.close()
        wall, _ = timed(build, args.reps, sync)
        base, _ = timed(upload_only, args.reps, sync)
        s3 = wl.scan_points[wl.scan_offsets[3]:wl.scan_offsets[4]]
        want = po.find(po.slice_params(**osp), s3, wl.map_points[:64], wl.x0[3])
        emit(name, fixed_points=int(wl.scan_offsets[-1]), wall_ms=wall, upload_alone_ms=base, build_and_query_ms=wall - base,
             parity_bit_identical=bool(np.array_equal(build.pairs, want)))
    for name, finder, osp in (
            ("a4 grid over the %d-point map + 1 query call" % args.map_points, api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3), dict(finder=po.FINDER_NN, max_distance=0.3)),
            ("a5 distance map over the %d-point map + 1 query call" % args.map_points, api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05), dict(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.05))):
        s3 = np.ascontiguousarray(wl.scan_points[wl.scan_offsets[3]:wl.scan_offsets[4]])
        x = synth.invert_poses(wl.x0[3:4].astype(np.float64))[0].astype(np.float32)

        def build():
            fresh = api.CloudSet(ctx, wl.map_points)
            finder.setFixed(fresh); finder.setMoving(s3); finder.setLocalMapInSensor(x); build.pairs = finder.compute(); fresh.close()

        def upload_only():
            fresh = api.CloudSet(ctx, wl.map_points); fresh.close()
        wall, _ = timed(build, args.reps, sync)
        base, _ = timed(upload_only, args.reps, sync)
        t0 = time.perf_counter(); want = po.find(po.slice_params(**osp), wl.map_points, s3, x); cpu_s = time.perf_counter() - t0
        emit(name, fixed_points=args.map_points, wall_ms=wall, upload_alone_ms=base, build_and_query_ms=wall - base, cpu_port_ms_1core=cpu_s * 1e3,
             pairs=int(len(want)), parity_bit_identical=bool(np.array_equal(build.pairs, want)))

    # ---- f1: clip a map-sized scene, merge a scan into it ---------------------------------------------------------------------------
    proj = api.PointNormal2fProjectorPolar(args.beams, -math.pi, math.pi, 0.3, 30.0)
    opr = po.Projector(args.beams, -math.pi, math.pi, 0.3, 30.0, 0.0)
    pose = synth.invert_poses(wl.x_true[5:6])[0].astype(np.float32)       # robot in local map
    clipper = api.SceneClipperProjective2D(ctx, proj, voxelize_resolution=0.0)
    clipped = api.CloudSet.reserved(ctx, args.beams + 64)
    clipper.setFullScene(mp); clipper.setClippedSceneInRobot(clipped); clipper.setRobotInLocalMap(pose)
    wall, best = timed(lambda: clipper.compute(), args.reps * 5, sync)
    got = clipped.download()
    t0 = time.perf_counter(); want = po.clip_scene(opr, wl.map_points, pose); cpu_s = time.perf_counter() - t0
    want_pts = want[0] if isinstance(want, tuple) else want
    emit("f1 clip %d-point scene" % args.map_points, wall_ms=wall, best_ms=best, points_per_s=args.map_points / (wall * 1e-3),
         algorithmic_GBs=(16.0 * args.map_points + 48.0 * args.beams) / (wall * 1e-3) / 1e9, cpu_port_ms_1core=cpu_s * 1e3, clipped=int(len(got)),
         parity_bit_identical=bool(np.array_equal(got, want_pts)))

    scan5 = np.ascontiguousarray(wl.scan_points[wl.scan_offsets[5]:wl.scan_offsets[6]])
    meas5 = api.CloudSet(ctx, scan5)
    x5 = pose                                                               # the measurement (sensor frame) in the scene
    scene = api.CloudSet.reserved(ctx, args.map_points + 64 * args.beams)
    merger = api.MergerProjective2D(ctx, proj, 0.2)
    merger.setScene(scene); merger.setMeasurement(meas5); merger.setMeasurementInScene(x5)

    def merge_once():
        scene.upload(wl.map_points); merger.compute()

    wall, _ = timed(merge_once, args.reps, sync)
    got = scene.download()
    t0 = time.perf_counter(); want, _ = po.merge_scene(opr, wl.map_points, scan5, x5, 0.2); cpu_s = time.perf_counter() - t0
    emit("f1 upload + merge one scan into a %d-point scene" % args.map_points, wall_ms=wall, cpu_port_ms_1core=cpu_s * 1e3, merged_size=int(len(got)),
         parity_bit_identical=bool(np.array_equal(got, want)))
    bad = [r["row"] for r in out if not r["parity_bit_identical"]]
    if bad:
        print("PARITY FAILED: " + "; ".join(bad), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
