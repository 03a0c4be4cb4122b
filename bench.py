#!/usr/bin/env python3
"""bench.py -- scan-to-map alignments/sec on MI355X (BASELINE.json metric), one process per GPU.

Workload (BASELINE.json configs[1]): a batch of 1000 synthetic 1081-beam scans against ONE 100k-point
local map, 20 Gauss-Newton iterations per alignment, reference role assignment (fixed = scan,
moving = map, projective finder -- SURVEY.md section 8d "role A / projective").  One step = one pass of the
batch through lsm2d_align_batch with the clouds already resident in HBM.  With N GPUs every rank
aligns its own 1000 scans (weak scaling) against the map broadcast from rank 0 over RCCL; there is
no data-path collective.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def algorithmic_bytes_per_alignment(role: str, finder: str, n_map: int, n_scan_mean: float, bins: int, iterations: int) -> float:
    """SURVEY.md section 8(d) table, per GN iteration (+ one-off term):
    A/projective  16*N_m (stream moving) + 48*Bins (canvas key write+read, fixed-cell read, winner gather) + 64; once 16*N_s
    A/nn          16*N_m + 28*C (C counted at N_m) + 64; once 16*N_s
    B/nn          16*N_s + N_s*(24*d + 8*L + 8) + 64 with L = 20, d = ceil(log2(N_m/L))"""
    if role == "A" and finder == "projective":
        return iterations * (16.0 * n_map + 48.0 * bins + 64.0) + 16.0 * n_scan_mean
    if role == "A" and finder == "nn":
        return iterations * (16.0 * n_map + 28.0 * n_map + 64.0) + 16.0 * n_scan_mean
    if role == "B" and finder == "nn":
        d = math.ceil(math.log2(n_map / 20.0))
        return iterations * (16.0 * n_scan_mean + n_scan_mean * (24.0 * d + 8.0 * 20 + 8.0) + 64.0)
    raise SystemExit("unsupported role/finder combination")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scans", type=int, default=1000, help="alignments per GPU per step")
    ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="alignments timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--role", choices=["A", "B"], default="A", help="A: fixed=scan, moving=map (reference tracker wiring); B: fixed=map, moving=scan")
    ap.add_argument("--finder", choices=["projective", "nn"], default="projective")
    ap.add_argument("--max-distance", type=float, default=0.5, help="NN finder gate [m]")
    ap.add_argument("--total-candidates", type=int, default=0,
                    help="BASELINE configs[3]: a fixed sweep of this many candidate alignments sharded over the ranks (strong scaling); "
                         "overrides --scans with this rank's share and gathers the poses on every rank at the end")
    ap.add_argument("--unique-scans", type=int, default=0, help="ray-cast only this many scans; candidates reuse them through an index array (loop-closure sweep)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or bool(os.environ.get("LSM2D_BENCH_FORCE_DIST"))   # the env var rehearses the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from srrg2_laser_slam_2d_amd import api, distributed, synth

    strong = args.total_candidates > 0
    if strong:
        lo, hi = distributed.shard_range(args.total_candidates, rank, world)
        args.scans = hi - lo

    # ---- inputs: the shared local map comes from rank 0 (RCCL broadcast), each rank ray-casts its own scans
    world_geom = synth.make_world(args.seed)
    map_dev = distributed.broadcast_map(
        synth.make_map(world_geom, args.map_points, seed=args.seed) if rank == 0 else None, args.map_points, local_rank)
    n_unique = args.unique_scans if 0 < args.unique_scans < args.scans else args.scans
    wl = synth.make_workload(n_unique, args.map_points, seed=args.seed, n_beams=args.beams, pose_seed_offset=rank,
                             world=world_geom, map_points=np.zeros((0, 4), np.float32))
    scan_index = None
    if n_unique < args.scans:      # candidate i = (scan i mod n_unique, its own perturbed initial guess)
        scan_index = (np.arange(args.scans) % n_unique).astype(np.int32)
        st = synth.Stream(args.seed + 1000 + rank, salt=9)
        delta = st.uniform(3 * args.scans, -0.05, 0.05).reshape(args.scans, 3)
        t_true = synth.invert_poses(wl.x_true)[scan_index]
        wl.x_true = wl.x_true[scan_index]
        wl.x0 = synth.invert_poses(synth.compose_poses(t_true, delta)).astype(np.float32)
    # a dedicated torch stream, made current: the kernels, the HIP events around them and torch's own view all sit on it
    # (the legacy default stream synchronises device-wide, which doubles the per-call latency of the single-scan config)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(side)
    ctx = api.Context(local_rank, stream=side.cuda_stream)
    map_set = api.CloudSet(ctx, map_dev)                      # stays in HBM, no host copy
    scan_set = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    if args.finder == "projective":
        proj = api.PointNormal2fProjectorPolar(args.beams, -np.pi, np.pi, 0.3, 30.0)
        finder = api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8)
    else:
        finder = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=args.max_distance, normal_cos=0.8)
    aligner = api.MultiAligner2D(ctx, max_iterations=args.iterations, min_num_inliers=10)
    aligner.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10))
    if args.role == "A":
        x0, x_true = wl.x0, wl.x_true
        idx = None if scan_index is None else scan_index[None, :]

        def step():
            return aligner.compute_batch([scan_set], [map_set], x0, fixed_index=idx)
    else:                           # the estimate is scan-in-map
        x0 = synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32); x_true = synth.invert_poses(wl.x_true)
        idx = None if scan_index is None else scan_index[None, :]

        def step():
            return aligner.compute_batch([map_set], [scan_set], x0, moving_index=idx)

    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        res = step()
        kernel_ms.append(res.kernel_ms)          # HIP events around the k_align launch, on the launch stream
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    if strong and use_dist:          # the sweep's consumer wants every candidate's pose: one all_gather of 12 B per candidate
        counts = [distributed.shard_range(args.total_candidates, r, world) for r in range(world)]
        pad = max(h - l for l, h in counts)
        mine = np.zeros((pad, 3), np.float32); mine[: len(res.pose)] = res.pose
        allp = distributed.gather_results(mine)
        assert allp.shape == (pad * world, 3)
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness gate: a timing only counts if the poses are right (noise-free data -> generating pose)
    # (the NN finder matches discrete map points ~N_m/220 m apart, so it lands within millimetres, not 1e-4)
    err = np.abs(res.pose - x_true)
    err[:, 2] = np.abs((err[:, 2] + np.pi) % (2 * np.pi) - np.pi)
    tol_m, tol_rad = (1e-4, 1e-4) if args.finder == "projective" else (5e-3, 2e-3)
    ok = bool(np.all(res.status == 0) and err[:, :2].max() < tol_m and err[:, 2].max() < tol_rad)
    if use_dist:
        f = torch.tensor([1 if ok else 0], device="cuda"); dist.all_reduce(f, op=dist.ReduceOp.MIN); ok = bool(f.item())

    if rank == 0:
        n_total = (args.total_candidates if strong else args.scans * world) * args.steps
        bytes_per_alignment = algorithmic_bytes_per_alignment(args.role, args.finder, args.map_points,
                                                              float(np.diff(wl.scan_offsets).mean()), args.beams, args.iterations)
        k_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_alignment * args.scans / (k_ms * 1e-3) / 1e9
        traffic = None; valu = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")     # written from the rocprofv3 --pmc passes, see profiles/README.md
        default_cfg = (args.role, args.finder, args.scans, args.map_points, args.iterations, args.beams) == ("A", "projective", 1000, 100000, 20, 1081)
        if os.path.exists(tpath) and default_cfg:
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("k_align_hbm_bytes_per_launch")
                vi = tj.get("k_align_valu_insts_per_launch")      # SQ_INSTS_VALU of the same launch: the limiter that matters here
                if vi:
                    visits = args.scans * args.iterations * (args.map_points + float(np.diff(wl.scan_offsets).mean())) / 64.0
                    valu = {"insts_per_launch": vi, "insts_per_point_visit": vi / visits, "source": "profiles/traffic.json (rocprofv3 --pmc SQ_INSTS_VALU)"}
            except Exception:
                traffic = None
        out = {
            "metric": "scan-to-map alignments/sec (1081-beam vs 100k-pt map, 20 GN iters)",
            "value": n_total / elapsed, "unit": "alignments/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s%d scans/GPU x %d-beam vs one %d-pt map, %d GN iters, role %s (%s), %s finder"
                                   % ("configs[1]: " if default_cfg else "", args.scans, args.beams, args.map_points, args.iterations, args.role,
                                      "fixed=scan, moving=map" if args.role == "A" else "fixed=map, moving=scan", args.finder),
                       "unique_scans": n_unique,
                       "alignments_per_gpu": args.scans, "map_points": args.map_points, "beams": args.beams,
                       "iterations": args.iterations, "parallelism": "alignments sharded, map replicated (RCCL broadcast)"},
            "parity_ok": ok, "max_pose_err_m": float(err[:, :2].max()), "max_pose_err_rad": float(err[:, 2].max()),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "valu": valu, "kernel": "k_align", "kernel_ms": k_ms,
                         "note": "achieved = SURVEY.md 8(d) algorithmic bytes / launch time; the map (<= 8 MB of xy) is L2 / Infinity-Cache "
                                 "resident, so achieved can exceed the HBM peak: the measured limiter is VALU issue (DESIGN.md section 5)",
                         "algorithmic_bytes_per_launch": bytes_per_alignment * args.scans},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle as po       # the checker, timed as the CPU baseline ("port")
            ns = min(args.cpu_sample, n_unique)
            offs = wl.scan_offsets[: ns + 1]
            map_host = map_dev.cpu().numpy()
            osp = po.slice_params(finder=po.FINDER_PROJECTIVE if args.finder == "projective" else po.FINDER_NN,
                                  canvas_cols=args.beams, max_distance=args.max_distance)
            po.lib()                                   # load (or build) the checker before the clock starts
            t1 = time.perf_counter()
            if args.role == "A":
                xo, _, st, _ = po.align_batch(po.aligner_params(args.iterations), osp, wl.scan_points[: offs[-1]], offs, map_host, x0[:ns], n_threads=1)
            else:
                xo = np.empty((ns, 3), np.float32)
                for i in range(ns):
                    r = po.align(po.aligner_params(args.iterations), [osp], [map_host], [wl.scan_points[offs[i]:offs[i + 1]]], x0[i])
                    xo[i] = r["pose"]
            cpu_s = time.perf_counter() - t1
            all_cores = None
            if args.role == "A" and ns >= 64:       # BASELINE.md section 3, second row: one alignment per host thread
                nt = max(1, len(os.sched_getaffinity(0)))
                t2 = time.perf_counter()
                po.align_batch(po.aligner_params(args.iterations), osp, wl.scan_points[: offs[-1]], offs, map_host, x0[:ns], n_threads=nt)
                all_cores = {"value": ns / (time.perf_counter() - t2), "cores": nt}
            d = np.abs(res.pose[:ns] - xo)
            # the same checker summing in the kernels' order must reproduce the device BIT FOR BIT (a handful of alignments)
            nbit = min(16, ns); bit_equal = 0
            if args.role == "A":
                for i in range(nbit):
                    rt = po.align(po.aligner_params(args.iterations, device_order=True), [osp], [wl.scan_points[offs[i]:offs[i + 1]]], [map_host], x0[i])
                    bit_equal += int(np.array_equal(res.pose[i], rt["pose"]) and np.array_equal(res.information[i], rt["H"]))
            else:
                for i in range(nbit):
                    rt = po.align(po.aligner_params(args.iterations, device_order=True), [osp], [map_host], [wl.scan_points[offs[i]:offs[i + 1]]], x0[i])
                    bit_equal += int(np.array_equal(res.pose[i], rt["pose"]) and np.array_equal(res.information[i], rt["H"]))
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "alignments/s", "cores": 1, "kind": "port",
                                   "sample": "first %d alignments of the same batch, CPU restatement of the reference algorithm (oracle/, gcc -O3 -march=native, fp32), %.1f s"
                                             % (ns, cpu_s),
                                   "max_pose_diff_gpu_vs_cpu_m": float(d[:, :2].max()), "max_pose_diff_gpu_vs_cpu_rad": float(d[:, 2].max()),
                                   "bit_identical_to_device_order_port": "%d of %d alignments (pose and information matrix)" % (bit_equal, nbit)}
            if all_cores:
                out["cpu_baseline"]["all_cores"] = all_cores
        print(json.dumps(out), flush=True)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
