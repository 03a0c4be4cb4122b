#!/usr/bin/env python3
"""Tracker-step latency through the C ABI alone: builds tests/cpp/track_step_bench.cpp with g++ against liblsm2d_hip.so, feeds
it one synthetic scene (MULTI.json parameters, as tests/bench/replay_bench.py) and prints its JSON lines (synchronous and
asynchronous calls) plus the CPU oracle's time for the same step.
    python tests/bench/track_step_bench.py [--steps 2000]"""
import argparse, json, math, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import synth


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--workdir", default=None, help="keep the built driver, its inputs and one command line per mode (cmd_<mode>.txt) here, e.g. to profile the driver itself")
    args = ap.parse_args()
    from oracle import pyoracle as po          # the checker: builds the reference local map and times the CPU step
    world = synth.make_world(5)
    S = [np.float32([0.2, 0.1, 0.1]), np.float32([-0.3, 0.0, math.pi])]
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    traj = [synth.sample_poses(world, 1, seed=21)[0]]
    for k in range(12):
        traj.append(synth.compose_poses(traj[-1][None, :], np.array([[0.05, 0.0, 0.02]]))[0])
    host_map = np.zeros((0, 4), np.float32)
    for k, t in enumerate(traj[:-1]):          # a local map as the tracker has it between key frames
        for s in S:
            sc = synth.make_scans(world, synth.compose_poses(np.array([t]), s[None, :].astype(np.float64)), n_beams=721, noise_sigma=0.01, seed=3 + k)[0]
            host_map, _ = po.merge_scene(opr, host_map, sc, np.float32(synth.compose_poses(t[None, :], s[None, :].astype(np.float64))[0]), 0.2)
    # the step's two LaserMessages as raw ranges; the "points in" modes get them preprocessed by the oracle (bit-identical to the device)
    a0, a1 = -2.34747, 2.35619
    ranges = [synth.make_scan_ranges(world, synth.compose_poses(np.array([traj[-1]]), s[None, :].astype(np.float64)), n_beams=721, angle_min=a0, angle_max=a1,
                                     noise_sigma=0.01, seed=99 + i)[0] for i, s in enumerate(S)]
    pp = po.Preprocessor(721, a0, a1, 0.3, 20.0, 0.3, 5, 0.02)
    scans = [po.preprocess_scan(pp, r) for r in ranges]
    guess = synth.compose_poses(traj[-1][None, :], np.array([[0.03, -0.02, 0.02]]))[0]
    import contextlib
    if args.workdir:
        os.makedirs(args.workdir, exist_ok=True)
    with (contextlib.nullcontext(args.workdir) if args.workdir else tempfile.TemporaryDirectory()) as d:
        exe = os.path.join(d, "track_step_bench"); lib = os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "lib")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "track_step_bench.cpp"),
                        "-L" + lib, "-llsm2d_hip_experiments" if os.environ.get("LSM2D_EXPERIMENTS", "0") not in ("", "0") else "-llsm2d_hip", "-Wl,-rpath," + lib, "-o", exe], check=True)
        host_map.tofile(os.path.join(d, "map.bin")); scans[0].tofile(os.path.join(d, "s0.bin")); scans[1].tofile(os.path.join(d, "s1.bin"))
        out = {"local_map_points": int(len(host_map)), "scan_points": [int(len(s)) for s in scans]}
        ranges[0].tofile(os.path.join(d, "r0.bin")); ranges[1].tofile(os.path.join(d, "r1.bin"))
        for mode, key in ((0, "c_abi_sync"), (1, "c_abi_async"), (2, "c_abi_async_ranges_in")):
            if args.workdir:
                open(os.path.join(d, "cmd_%d.txt" % mode), "w").write(" ".join(
                    [exe, os.path.join(d, "map.bin"), os.path.join(d, "s0.bin"), os.path.join(d, "s1.bin"), repr(float(guess[0])), repr(float(guess[1])),
                     repr(float(guess[2])), "200", str(mode), os.path.join(d, "r0.bin"), os.path.join(d, "r1.bin"), repr(a0), repr(a1)]) + "\n")
            r = subprocess.run([exe, os.path.join(d, "map.bin"), os.path.join(d, "s0.bin"), os.path.join(d, "s1.bin"),
                                repr(float(guess[0])), repr(float(guess[1])), repr(float(guess[2])), str(args.steps), str(mode),
                                os.path.join(d, "r0.bin"), os.path.join(d, "r1.bin"), repr(a0), repr(a1)],
                               check=True, capture_output=True, text=True, timeout=300)
            if os.environ.get("LSM2D_TSB_DUMP"):          # debug builds (-DLSM2D_PHASE_CLOCKS) print from the kernel
                sys.stderr.write(r.stdout)
            out[key] = json.loads(r.stdout.strip().splitlines()[-1])
        # the same asynchronous step with the point-query finders in both slices: CorrespondenceFinderKDTree2D as the reference runs it (its tree REBUILT for
        # every new scan, correspondence_finder_kd_tree_2d.cpp:6-8,31-38: since round 4 one launch per build), the exact grid search, the distance map
        for fk in ("kdtree", "nn", "distmap"):
            r = subprocess.run([exe, os.path.join(d, "map.bin"), os.path.join(d, "s0.bin"), os.path.join(d, "s1.bin"),
                                repr(float(guess[0])), repr(float(guess[1])), repr(float(guess[2])), str(max(args.steps // 4, 50)), "1",
                                os.path.join(d, "r0.bin"), os.path.join(d, "r1.bin"), repr(a0), repr(a1)],
                               check=True, capture_output=True, text=True, timeout=300, env=dict(os.environ, LSM2D_TSB_FINDER=fk))
            out["c_abi_async_finder_" + fk] = json.loads(r.stdout.strip().splitlines()[-1])
        if os.environ.get("LSM2D_EXPERIMENTS", "0") not in ("", "0"):      # ("kd_wg_max_points" is an A/B knob of the experiments build: the shipped library builds a scan's tree in one launch)
            r = subprocess.run([exe, os.path.join(d, "map.bin"), os.path.join(d, "s0.bin"), os.path.join(d, "s1.bin"),
                                repr(float(guess[0])), repr(float(guess[1])), repr(float(guess[2])), str(max(args.steps // 4, 50)), "1",
                                os.path.join(d, "r0.bin"), os.path.join(d, "r1.bin"), repr(a0), repr(a1)],
                               check=True, capture_output=True, text=True, timeout=300, env=dict(os.environ, LSM2D_TSB_FINDER="kdtree", LSM2D_TSB_OPTIONS="kd_wg_max_points=0"))
            out["c_abi_async_finder_kdtree_level_loop_build"] = json.loads(r.stdout.strip().splitlines()[-1])
    # the same step on the CPU oracle
    osl = [po.slice_params(canvas_cols=721, range_max=20.0, normal_cos=0.9, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.01, min_num_correspondences=5, sensor_in_robot=tuple(S[0])),
           po.slice_params(canvas_cols=721, range_max=20.0, normal_cos=0.8, min_num_correspondences=5, sensor_in_robot=tuple(S[1]))]
    omega = np.diag([100.0, 100.0, 100.0]).astype(np.float32)
    reps = 200; t0 = time.perf_counter()
    for _ in range(reps):
        _ = [po.preprocess_scan(pp, r) for r in ranges]
    out["cpu_oracle_preprocess_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        g32 = guess.astype(np.float32)
        oclip, _ = po.clip_scene(opr, host_map, g32, S[0])
        r = po.align(po.aligner_params(10, prior_z=[0, 0, 0], prior_omega=omega), osl, scans, [oclip, oclip], np.zeros(3, np.float32))
        est = synth.compose_poses(guess[None, :], synth.invert_poses(r["pose"][None, :].astype(np.float64)))[0]
        hm = host_map
        for sc, s in zip(scans, S):
            hm, _ = po.merge_scene(opr, hm, sc, np.float32(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0]), 0.2)
    out["cpu_oracle_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / reps        # points in; add the preprocess line for ranges in
    out["pose_diff_gpu_vs_cpu"] = [float(abs(a - b)) for a, b in zip(out["c_abi_async"]["est_on_fresh_map"], est)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
