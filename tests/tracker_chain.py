"""A short live-tracker chain (raw ranges in -> preprocess -> clip -> align with two laser slices and an odometry prior -> merge,
MULTI.json parameters) run two ways -- on the CPU oracle and through the HIP path -- and reduced to per-step digests, so that both
can be held against ONE committed file (tests/golden/tracker_chain.json, written by tests/golden/make_tracker_chain.py from the oracle).

The fp32 path has no libm call except the beam directions of the preprocessor (cosf / sinf on the host, in the oracle and in the
library alike), so the digests are the same bits on every host with this image's glibc.
"""
import hashlib
import math

import numpy as np

from srrg2_laser_slam_2d_amd import synth

A0, A1 = -2.34747, 2.35619
S = [np.float32([0.2, 0.1, 0.1]), np.float32([-0.3, 0.0, math.pi])]
N_BEAMS, COLS, RMIN, RMAX = 721, 721, 0.3, 20.0
OMEGA = np.diag([100.0, 100.0, 100.0]).astype(np.float32)
ITS = 10


def digest(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:20]


def scenario(steps: int = 8, seed: int = 4):
    """trajectory, raw ranges of both scanners per pose, noisy odometry"""
    world = synth.make_world(seed)
    st = synth.Stream(seed, salt=11)
    traj = [synth.sample_poses(world, 1, seed=seed + 3)[0]]
    while len(traj) < steps + 1:
        nxt = synth.compose_poses(traj[-1][None, :], st.uniform(3, -0.05, 0.05)[None, :])[0]
        if synth._free(world, nxt[None, :2], 0.8)[0]:
            traj.append(nxt)
    traj = np.array(traj)
    ranges = [synth.make_scan_ranges(world, synth.compose_poses(traj, np.tile(s.astype(np.float64), (len(traj), 1))), n_beams=N_BEAMS,
                                     angle_min=A0, angle_max=A1, noise_sigma=0.01, seed=1 + i) for i, s in enumerate(S)]
    odo = [synth.compose_poses(synth.invert_poses(traj[k - 1:k]), traj[k:k + 1])[0] + st.uniform(3, -0.01, 0.01) for k in range(1, steps + 1)]
    return traj, ranges, odo


def _sensor_pose(est, s):
    return np.float32(synth.compose_poses(np.asarray(est, np.float64)[None, :], s[None, :].astype(np.float64))[0])


def run_oracle(po, steps: int = 8, record_every: int = 1):
    """record_every > 1: digests only of every record_every-th step and of the last one (the 1000-step replay of BASELINE configs[2])"""
    traj, ranges, odo = scenario(steps)
    pp = po.Preprocessor(N_BEAMS, A0, A1, RMIN, RMAX, 0.3, 5, 0.02)
    opr = po.Projector(COLS, -math.pi, math.pi, RMIN, RMAX, 0.0)
    osl = [po.slice_params(canvas_cols=COLS, range_max=RMAX, normal_cos=0.9, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.01,
                           min_num_correspondences=5, sensor_in_robot=tuple(S[0])),
           po.slice_params(canvas_cols=COLS, range_max=RMAX, normal_cos=0.8, min_num_correspondences=5, sensor_in_robot=tuple(S[1]))]
    host_map = np.zeros((0, 4), np.float32)
    for i, s in enumerate(S):
        host_map, _ = po.merge_scene(opr, host_map, po.preprocess_scan(pp, ranges[i][0]), _sensor_pose(traj[0], s), 0.2)
    est = traj[0].copy(); out = []
    for k in range(1, steps + 1):
        meas = [po.preprocess_scan(pp, ranges[i][k]) for i in range(2)]
        guess = synth.compose_poses(est[None, :], odo[k - 1][None, :])[0].astype(np.float32)
        clip, _ = po.clip_scene(opr, host_map, guess, S[0])
        r = po.align(po.aligner_params(ITS, prior_z=[0, 0, 0], prior_omega=OMEGA, device_order=True), osl, meas, [clip, clip], np.zeros(3, np.float32))
        est = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(np.asarray(r["pose"], np.float64)[None, :]))[0]
        for i, s in enumerate(S):
            host_map, _ = po.merge_scene(opr, host_map, meas[i], _sensor_pose(est, s), 0.2)
        if k % record_every and k != steps:
            continue
        out.append({"step": k, "scans": [digest(m) for m in meas], "scan_points": [int(len(m)) for m in meas], "clip_points": int(len(clip)), "clip": digest(clip),
                    "status": int(r["status"]), "pose_hex": [float(v).hex() for v in np.asarray(r["pose"], np.float32)],
                    "information": digest(np.asarray(r["H"], np.float32)), "map_points": int(len(host_map)), "map": digest(host_map)})
    return out


def run_device(api, ctx, steps: int = 8, record_every: int = 1, map_capacity: int = 50000):
    """the same chain through the C ABI's Python mirror: ranges in, everything else stays on the device (asynchronous clip / merge);
    between recorded steps nothing but the aligner's pose comes back to the host"""
    traj, ranges, odo = scenario(steps)
    proj = api.PointNormal2fProjectorPolar(COLS, -math.pi, math.pi, RMIN, RMAX)
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=RMIN, range_max=RMAX, voxelize_resolution=0.02)
    sets = [api.CloudSet.reserved(ctx, 1024), api.CloudSet.reserved(ctx, 1024)]
    local_map = api.CloudSet.reserved(ctx, map_capacity)
    clipper = api.SceneClipperProjective2D(ctx, proj, asynchronous=True, voxelize_resolution=0.0); clipper.setFullScene(local_map)
    merger = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=True); merger.setScene(local_map)
    al = api.MultiAligner2D(ctx, max_iterations=ITS, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.9), sensor_in_robot=S[0], robustifier=api.RobustifierCauchy(0.01),
        min_num_correspondences=5, fixed_slice_name="points_0", moving_slice_name="points"))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=S[1], min_num_correspondences=5,
        fixed_slice_name="points_1", moving_slice_name="points"))

    def measure(k):
        for i in range(2):
            pre.setRawData(ranges[i][k], A0, A1, 0.0, 30.0); pre.compute_into(sets[i])

    measure(0)
    for i, s in enumerate(S):
        merger.setMeasurement(sets[i]); merger.setMeasurementInScene(_sensor_pose(traj[0], s)); merger.compute()
    est = traj[0].copy(); out = []
    for k in range(1, steps + 1):
        measure(k)
        guess = synth.compose_poses(est[None, :], odo[k - 1][None, :])[0].astype(np.float32)
        clipper.setRobotInLocalMap(guess); clipper.setSensorInRobot(S[0])
        clipped = clipper.compute()
        al.setFixed({"points_0": sets[0], "points_1": sets[1]}); al.setMoving({"points": clipped}); al.setMovingInFixed([0, 0, 0]); al.setPrior([0, 0, 0], OMEGA)
        status = al.compute()
        x = al.movingInFixed()
        est = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(x[None, :].astype(np.float64)))[0]
        rec = {"step": k, "status": int(status), "pose_hex": [float(v).hex() for v in x], "information": digest(al.informationMatrix().astype(np.float32))}
        recorded = not (k % record_every and k != steps)
        if recorded:
            meas = [s_.download() for s_ in sets]; clip = clipped.download()          # read back for the digests only (after the aligner)
        for i, s in enumerate(S):
            merger.setMeasurement(sets[i]); merger.setMeasurementInScene(_sensor_pose(est, s)); merger.compute()
        if not recorded:
            continue
        m = local_map.download()
        rec.update({"scans": [digest(v) for v in meas], "scan_points": [int(len(v)) for v in meas], "clip_points": int(len(clip)), "clip": digest(clip),
                    "map_points": int(len(m)), "map": digest(m)})
        out.append(rec)
    return out
