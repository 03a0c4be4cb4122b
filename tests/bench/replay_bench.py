#!/usr/bin/env python3
"""BASELINE configs[2]: stage_segway_double_config_MULTI-style replay -- per-scan tracker alignment on one MI355X,
pose vs the CPU oracle.  The bag (segway_double_3.bag) is not available offline, so the replay is synthetic with
the MULTI parameters: 721-column projectors over [-pi, pi], range 0.3-20 m, two laser slices with
their own extrinsics (laser_0: Cauchy 0.01, normal_cos 0.9; laser_1: no robustifier, normal_cos 0.8), odometry prior,
10 iterations, clipper before / merger after the aligner, local map kept on the device.

Every step runs the SAME three calls on both sides (clip -> align -> merge); the CPU side is the oracle (checker).
    python tests/bench/replay_bench.py [--steps 200]
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scan-noise", type=float, default=0.01)
    ap.add_argument("--sync-calls", action="store_true",
                    help="every clip / merge call waits for its result (5 host synchronisations per step); default: the clipper and the "
                         "merger only queue their work and the step synchronises once, for the aligner's pose")
    ap.add_argument("--sequential-oracle", action="store_true",
                    help="the CPU side sums H, b pair after pair (the reference's order: poses then agree to ~1e-7 per step); default: it "
                         "sums in the kernels' order (lsmo_aligner_params.device_order), and the two pipelines must stay BIT-IDENTICAL")
    ap.add_argument("--kernel-timing", action="store_true",
                    help="record HIP events around every launch to report the aligner's kernel time (costs ~30 us per step; default off)")
    ap.add_argument("--chained", action="store_true",
                    help="let the GPU pipeline run on its own state for the whole trajectory (reports drift); default is lockstep: "
                         "before every step the GPU state is reset to the CPU state, so differences are per-step")
    args = ap.parse_args()
    from oracle import pyoracle as po
    from srrg2_laser_slam_2d_amd import api, synth

    world = synth.make_world(args.seed)
    S0, S1 = np.float32([0.2, 0.1, 0.1]), np.float32([-0.3, 0.0, math.pi])
    # random walk in free space: U(-0.05, 0.05) per step as synthetic_scene_generator.cpp:167-178
    st = synth.Stream(args.seed, salt=11)
    robot = synth.sample_poses(world, 1, seed=args.seed + 3)[0]
    traj = [robot]
    while len(traj) < args.steps + 1:
        d = st.uniform(3, -0.05, 0.05)
        nxt = synth.compose_poses(traj[-1][None, :], d[None, :])[0]
        if synth._free(world, nxt[None, :2], 0.8)[0]:
            traj.append(nxt)
    traj = np.array(traj)
    sensors0 = synth.compose_poses(traj, np.tile(S0.astype(np.float64), (len(traj), 1)))
    sensors1 = synth.compose_poses(traj, np.tile(S1.astype(np.float64), (len(traj), 1)))
    sc0, of0 = synth.make_scans(world, sensors0, n_beams=721, range_max=20.0, noise_sigma=args.scan_noise, seed=1)
    sc1, of1 = synth.make_scans(world, sensors1, n_beams=721, range_max=20.0, noise_sigma=args.scan_noise, seed=2)
    odo_noise = st.uniform(3 * len(traj), -0.01, 0.01).reshape(-1, 3)

    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    ctx = api.Context(0, kernel_timing=args.kernel_timing)
    local_map = api.CloudSet.reserved(ctx, 400000)
    clipper = api.SceneClipperProjective2D(ctx, proj, asynchronous=not args.sync_calls, voxelize_resolution=0.0); clipper.setFullScene(local_map)
    merger = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=not args.sync_calls); merger.setScene(local_map)
    al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.9), sensor_in_robot=S0, robustifier=api.RobustifierCauchy(0.01),
        min_num_correspondences=5, fixed_slice_name="points_0", moving_slice_name="points"))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=S1, min_num_correspondences=5,
        fixed_slice_name="points_1", moving_slice_name="points"))
    osl = [po.slice_params(canvas_cols=721, range_max=20.0, normal_cos=0.9, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.01,
                           min_num_correspondences=5, sensor_in_robot=tuple(S0)),
           po.slice_params(canvas_cols=721, range_max=20.0, normal_cos=0.8, min_num_correspondences=5, sensor_in_robot=tuple(S1))]
    omega = np.diag([100.0, 100.0, 100.0]).astype(np.float32)      # odometry prior information

    # both pipelines start from the first pair of scans merged at the true start pose
    first0, first1 = sc0[of0[0]:of0[1]], sc1[of1[0]:of1[1]]
    host_map = np.zeros((0, 4), np.float32)
    for meas, sp in ((first0, sensors0[0]), (first1, sensors1[0])):
        host_map, _ = po.merge_scene(opr, host_map, meas, np.float32(sp), 0.2)
    local_map.upload(host_map)
    est_gpu = traj[0].copy(); est_cpu = traj[0].copy()
    t_gpu = t_cpu = 0.0; gpu_kernel_ms = 0.0
    phase = {"clip": 0.0, "upload_scans": 0.0, "align": 0.0, "merge": 0.0}
    scan_sets = [api.CloudSet.reserved(ctx, 1024), api.CloudSet.reserved(ctx, 1024)]     # reused every step: no allocation per scan
    max_dp = max_dth = 0.0; err_truth = []
    for k in range(1, args.steps + 1):
        a0, a1 = sc0[of0[k]:of0[k + 1]], sc1[of1[k]:of1[k + 1]]
        odo = synth.compose_poses(synth.invert_poses(traj[k - 1:k]), traj[k:k + 1])[0] + odo_noise[k]      # noisy relative motion
        if not args.chained:
            local_map.upload(host_map); est_gpu = est_cpu.copy()        # not timed: parity bookkeeping only
        # ---- GPU: clip -> align (2 laser slices + prior) -> merge, local map stays on the device
        t = time.perf_counter()
        guess = synth.compose_poses(est_gpu[None, :], odo[None, :])[0].astype(np.float32)
        clipper.setRobotInLocalMap(guess); clipper.setSensorInRobot(S0)
        clipped = clipper.compute()
        t1 = time.perf_counter(); phase["clip"] += t1 - t
        scan_sets[0].upload(a0); scan_sets[1].upload(a1)
        t2 = time.perf_counter(); phase["upload_scans"] += t2 - t1
        al.setFixed({"points_0": scan_sets[0], "points_1": scan_sets[1]}); al.setMoving({"points": clipped}); al.setMovingInFixed([0, 0, 0])
        al.setPrior([0, 0, 0], omega)
        status = al.compute(); gpu_kernel_ms += al._result.kernel_ms
        x_gpu = al.movingInFixed().astype(np.float64)
        est_gpu = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(x_gpu[None, :]))[0]
        t3 = time.perf_counter(); phase["align"] += t3 - t2
        for sset, S in ((scan_sets[0], S0), (scan_sets[1], S1)):
            merger.setMeasurement(sset); merger.setMeasurementInScene(synth.compose_poses(est_gpu[None, :], S[None, :].astype(np.float64))[0])
            merger.compute()
        t4 = time.perf_counter(); phase["merge"] += t4 - t3
        t_gpu += t4 - t
        # ---- CPU oracle: the same calls
        t = time.perf_counter()
        guess_c = synth.compose_poses(est_cpu[None, :], odo[None, :])[0].astype(np.float32)
        oclip, _ = po.clip_scene(opr, host_map, guess_c, S0)
        r = po.align(po.aligner_params(10, prior_z=[0, 0, 0], prior_omega=omega, device_order=not args.sequential_oracle), osl, [a0, a1], [oclip, oclip], np.zeros(3, np.float32))
        est_cpu = synth.compose_poses(guess_c[None, :].astype(np.float64), synth.invert_poses(r["pose"][None, :].astype(np.float64)))[0]
        for meas, S in ((a0, S0), (a1, S1)):
            host_map, _ = po.merge_scene(opr, host_map, meas, np.float32(synth.compose_poses(est_cpu[None, :], S[None, :].astype(np.float64))[0]), 0.2)
        t_cpu += time.perf_counter() - t
        dp = np.abs(est_gpu - est_cpu)
        max_dp = max(max_dp, dp[:2].max()); max_dth = max(max_dth, abs((dp[2] + math.pi) % (2 * math.pi) - math.pi))
        err_truth.append(np.abs(est_gpu - traj[k])[:2].max())
        assert status == r["status"], (k, status, r["status"])
    out = {"mode": "chained" if args.chained else "lockstep (GPU state reset to the CPU state before every step)",
           "calls": "synchronous" if args.sync_calls else "asynchronous clip / upload / merge, one synchronisation per step",
           "config": "configs[2]: synthetic MULTI-parameter replay (721 cols, 10 its, 2 laser slices + odometry prior, clip+merge)",
           "gpu_phase_ms_per_step": {k: 1e3 * v / args.steps for k, v in phase.items()},
           "steps": args.steps, "gpu_ms_per_step_wall": 1e3 * t_gpu / args.steps, "gpu_align_kernel_ms_per_step": gpu_kernel_ms / args.steps,
           "cpu_oracle_ms_per_step_wall": 1e3 * t_cpu / args.steps, "max_pose_diff_gpu_vs_cpu_m": float(max_dp), "max_pose_diff_gpu_vs_cpu_rad": float(max_dth),
           "oracle_summation": "sequential (reference order)" if args.sequential_oracle else "device order",
           "final_map_points_gpu": int(local_map.n_points), "final_map_points_cpu": int(len(host_map)),
           "final_maps_bit_identical": bool(np.array_equal(local_map.download(), host_map)),
           "max_abs_translation_error_vs_truth_m": float(max(err_truth)), "final_translation_error_vs_truth_m": float(err_truth[-1])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
