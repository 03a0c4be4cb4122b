#!/usr/bin/env python3
"""Throughput against BATCH SIZE on configs[1]'s geometry (1081-beam scans against one 100k-point map, 20 GN iterations, role A, projective finder), with fresh start
poses every step -- VERDICT r5 item 2: one workgroup per alignment and 1024 resident workgroup slots mean a batch of 1025 .. 1100 alignments starts a second,
nearly empty dispatch round; is there a cliff?  (Round 6: such batches run in narrow workgroups or PACKED -- `--options align_width=512|256|1024` forces a form.)

    python tools/batch_size_sweep.py [--sizes 128,256,...] [--out gpurun_out/batch_size_sweep.jsonl]

One JSON line per size: alignments/s and ms per step (wall, prepare_batch + set_init_poses + lsm2d_align_batch per step, new poses every step), the k_align launch's
duration by HIP events and the clock the chip held inside it (a second, timed loop), whether the placement's estimate ran.  Product API only (no oracle): the gate is
the generating pose within 1e-4 m / 1e-4 rad (noise-free data)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="128,256,512,768,1000,1024,1025,1100,1536,2048,3000,4096,8192")
    ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--pose-sets", type=int, default=4)
    ap.add_argument("--seconds", type=float, default=0.6, help="timed region per size (at least 20 steps)")
    ap.add_argument("--options", default="", help="context options, key=value,key=value (e.g. sum_order=1)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    from srrg2_laser_slam_2d_amd import api, synth
    sizes = [int(v) for v in args.sizes.split(",")]
    wl = synth.make_workload(max(8192, max(sizes)), args.map_points, seed=0, n_beams=args.beams)      # (always the same 8192 scans: the sampled poses depend on how many are drawn)
    ctx = api.Context(0, kernel_timing=False)
    opts = {}
    for kv in filter(None, args.options.split(",")):
        k, _, v = kv.partition("="); ctx.set_option(k.strip(), int(v)); opts[k.strip()] = int(v)
    proj = api.PointNormal2fProjectorPolar(args.beams, -np.pi, np.pi, 0.3, 30.0)
    al = api.MultiAligner2D(ctx, max_iterations=args.iterations, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=10))
    themap = api.CloudSet(ctx, wl.map_points)
    t_true = synth.invert_poses(wl.x_true)
    sets_all = [wl.x0.astype(np.float32)]
    for k in range(1, args.pose_sets):
        dk = synth.Stream(104729 * k, salt=6).uniform(3 * len(t_true), -0.05, 0.05).reshape(len(t_true), 3)
        sets_all.append(synth.invert_poses(synth.compose_poses(t_true, dk)).astype(np.float32))
    rows = []
    for n in sizes:
        scans = api.CloudSet(ctx, wl.scan_points[: wl.scan_offsets[n]], wl.scan_offsets[: n + 1])
        sets = [x[:n] for x in sets_all]
        prep = al.prepare_batch([scans], [themap], sets[0])
        ok = True
        for k in range(len(sets)):      # gate every pose set
            prep.set_init_poses(sets[k]); r = prep.run()
            e = np.abs(r.pose - wl.x_true[:n]); e[:, 2] = np.abs((e[:, 2] + np.pi) % (2 * np.pi) - np.pi)
            ok = ok and bool(np.all(r.status == 0) and e[:, :2].max() < 1e-4 and e[:, 2].max() < 1e-4)
        t_end = time.perf_counter() + 0.3      # clock ramp
        i = 0
        while time.perf_counter() < t_end:
            prep.set_init_poses(sets[i % len(sets)]); prep.run(); i += 1
        steps = 0; t0 = time.perf_counter()
        while steps < 20 or time.perf_counter() - t0 < args.seconds:
            prep.set_init_poses(sets[steps % len(sets)]); prep.run(); steps += 1
        wall = time.perf_counter() - t0
        est = bool(ctx.get_option("last_cull_estimate"))
        ctx.set_option("kernel_timing", 1)
        km, ck = [], []
        for j in range(24):
            prep.set_init_poses(sets[j % len(sets)]); r = prep.run(); km.append(r.kernel_ms); ck.append(r.kernel_clock_mhz)
        ctx.set_option("kernel_timing", 0)
        row = {"n": n, "alignments_per_s": n * steps / wall, "ms_per_step": wall / steps * 1e3, "us_per_alignment": wall / steps / n * 1e6, "steps": steps,
               "kernel_ms": float(np.mean(km[4:])), "clock_mhz_in_kernel": float(np.median([c for c in ck if c > 0])) if any(c > 0 for c in ck) else None,
               "estimate_launched": est, "last_align_path": ctx.get_option("last_align_path"), "parity_ok": ok, "options": opts}
        rows.append(row)
        print(json.dumps(row), flush=True)
        scans.close()
    if args.out:
        with open(args.out, "w") as f:
            for row in rows:
                f.write(json.dumps(row) + "\n")
    ctx.close()


if __name__ == "__main__":
    main()
