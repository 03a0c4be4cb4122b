"""GPU parity tests, rows f1-f2: scene clipper, merger, raw-data preprocessor and the tracker step built from them, bit-exact.

The HIP path (through the C ABI) against the CPU oracle on identical inputs.  Bars (BASELINE.json north_star): index work bit-exact; H / b / chi 2e-5 relative
against the fp64 oracle and BITWISE against the fp32 oracle in the launch's order; aligner pose within 1e-4 m / 1e-4 rad (gpu_helpers.POSE_TOL_*)."""
import json
import math

import numpy as np
import pytest

import fuzz_cases
from conftest import golden_path, has_experiments, need_experiments, xset
from gpu_helpers import (_same_correspondence_sets, _assert_bitwise_equal_to_device_order_oracle, _pose_diff, _Envelope, _projector, _aligner, _nn_aligner, _ranges_in_pose_out_step, _kd_finder, _kd_aligner, _neg_eps, _oracle_slice, POSE_TOL_M, POSE_TOL_RAD)
from srrg2_laser_slam_2d_amd import api, synth

pytestmark = pytest.mark.gpu


# ---- mapping around the aligner (row f1): clipper, merger, device-resident local map ------------------------
def test_scene_clipper_bit_exact(ctx, po):
    world = synth.make_world(3)
    m = synth.make_map(world, 60000)
    robots = synth.sample_poses(world, 4, seed=5)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    scene = api.CloudSet(ctx, m)
    clipper = api.SceneClipperProjective2D(ctx, proj, voxelize_resolution=0.0)
    clipper.setFullScene(scene)
    for robot, S in zip(robots, ([0, 0, 0], [0.2, 0.1, 0.1], [-0.3, 0.0, math.pi], [0, 0, 0])):
        clipper.setRobotInLocalMap(robot); clipper.setSensorInRobot(S)
        clipped = clipper.compute()
        want, wsrc = po.clip_scene(po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0),
                                   m, np.float32(robot), np.float32(S))
        assert 300 < len(want) <= 721
        assert np.array_equal(clipper.source_indices, wsrc)
        assert np.array_equal(clipped.download(), want)
    with pytest.raises(RuntimeError):
        api.SceneClipperProjective2D(ctx, proj, voxelize_resolution=0.0).compute()          # missing scene (scene_clipper_projective_2d.cpp:12-17)


def test_merger_bit_exact_and_grows_in_place(ctx, po):
    world = synth.make_world(3)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    poses = synth.sample_poses(world, 6, seed=8)
    scans, offs = synth.make_scans(world, poses, n_beams=721, noise_sigma=0.01, seed=3)
    scene = api.CloudSet.reserved(ctx, 20000)
    host_scene = scans[offs[0]:offs[1]].copy()            # first scan seeds the local map at the origin of ITS frame
    # express everything in the frame of the first sensor pose: measurement_in_scene = T0^-1 * Ti
    scene.upload(host_scene)
    merger = api.MergerProjective2D(ctx, proj, merge_threshold=0.2)
    merger.setScene(scene)
    t0_inv = synth.invert_poses(poses[:1])
    for i in range(1, 6):
        meas = scans[offs[i]:offs[i + 1]]
        mis = synth.compose_poses(t0_inv, poses[i:i + 1])[0].astype(np.float32)
        merger.setMeasurement(meas); merger.setMeasurementInScene(mis)
        n = merger.compute()
        host_scene, counts = po.merge_scene(opr, host_scene, meas, mis, 0.2)
        assert n == len(host_scene) and merger.counts == counts
        assert np.array_equal(scene.download(), host_scene)
    assert len(host_scene) > offs[1] - offs[0]
    # capacity is enforced, never overrun
    small = api.CloudSet.reserved(ctx, 800); small.upload(scans[offs[0]:offs[1]][:200])
    merger.setScene(small)
    with pytest.raises(api.Lsm2dError):
        merger.compute()


def test_merging_several_measurements_in_one_call(ctx, po):
    """lsm2d_merge_scenes: n measurements, each at its own pose, merged in order by ONE launch -- bit for bit what n lsm2d_merge_scene
    calls (and the oracle) give, synchronous and asynchronous, with sizes the device alone knows, and falling back to single calls for
    large scenes."""
    world = synth.make_world(5)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    robots = synth.sample_poses(world, 3, seed=9)
    scans = [synth.make_scans(world, robots[i:i + 1], n_beams=721, noise_sigma=0.01, seed=4 + i)[0] for i in range(3)]
    poses = [np.float32(r) for r in robots]
    for n_map in (0, 1500, 40000):                                   # empty scene, tracker-sized, large (multi-launch path: one by one)
        base = synth.make_map(world, n_map, noise_sigma=0.01, seed=1) if n_map else np.zeros((0, 4), np.float32)
        want = base; want_counts = []
        for sc, p in zip(scans, poses):
            want, c = po.merge_scene(opr, want, sc, p, 0.2); want_counts.append(tuple(int(v) for v in c))
        for asynchronous in (False, True):
            for quiet in (False, True):                              # with timing events every merge is a launch of its own
                cx = api.Context(0, kernel_timing=False) if quiet else ctx
                try:
                    scene = api.CloudSet.reserved(cx, 60000); scene.upload(base)
                    mg = api.MergerProjective2D(cx, proj, 0.2, asynchronous=asynchronous); mg.setScene(scene)
                    sets = [api.CloudSet.reserved(cx, 1024) for _ in scans]
                    for st, sc in zip(sets, scans):
                        st.upload(sc)                                # left to the merge to unpack
                    size = mg.compute_all(sets, poses)
                    if not asynchronous:
                        assert size == len(want) and [tuple(c) for c in mg.counts] == want_counts
                    assert scene.n_points == len(want) and np.array_equal(scene.download(), want), (n_map, asynchronous, quiet)
                    # a second round on sizes the device alone knows (asynchronous) gives what the oracle gives from `want`
                    size2 = mg.compute_all(sets[:2], [poses[1], poses[0]])
                    want2 = want
                    for sc, p in ((scans[0], poses[1]), (scans[1], poses[0])):
                        want2, _ = po.merge_scene(opr, want2, sc, p, 0.2)
                    assert np.array_equal(scene.download(), want2) and (asynchronous or size2 == len(want2))
                finally:
                    if quiet:
                        cx.close()
    # clouds picked out of ONE multi-cloud set by index (a batch of scans as lsm2d_preprocess_scans returns it), in another order
    pts = np.concatenate(scans); offs = np.cumsum([0] + [len(s) for s in scans]).astype(np.int32)
    batch = api.CloudSet(ctx, pts, offs)
    base = synth.make_map(world, 1500, noise_sigma=0.01, seed=1)
    scene = api.CloudSet.reserved(ctx, 60000); scene.upload(base)
    mg = api.MergerProjective2D(ctx, proj, 0.2); mg.setScene(scene)
    size = mg.compute_all([batch, batch, batch], [poses[2], poses[0], poses[1]], indices=[2, 0, 1])
    want = base
    for i in (2, 0, 1):
        want, _ = po.merge_scene(opr, want, scans[i], poses[i], 0.2)
    assert size == len(want) and np.array_equal(scene.download(), want)
    with pytest.raises(Exception):
        mg.compute_all([batch], [poses[0]], indices=[3])
    with pytest.raises(Exception):
        api.MergerProjective2D(ctx, proj, 0.2).compute_all([], [])


def test_tracker_step_clip_align_merge_device_resident(ctx, po):
    """One tracker step as in apps/visual_test_tracker_2d.cpp:167-183 (clip -> align -> merge) with the local map kept
    on the device, against the same three steps of the oracle; MULTI-like wiring: two laser slices with extrinsics."""
    world = synth.make_world(5)
    m = synth.make_map(world, 40000, noise_sigma=0.005, seed=1)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    robot = synth.sample_poses(world, 1, seed=21)
    S0, S1 = np.float32([0.2, 0.1, 0.1]), np.float32([-0.3, 0.0, math.pi])
    scans = [synth.make_scans(world, synth.compose_poses(robot, S[None, :].astype(np.float64)), n_beams=721)[0] for S in (S0, S1)]
    guess = synth.compose_poses(robot, np.array([[0.03, -0.02, 0.02]]))[0].astype(np.float32)      # odometry-predicted robot pose
    # --- device pipeline
    local_map = api.CloudSet.reserved(ctx, 60000); local_map.upload(m)
    clipper = api.SceneClipperProjective2D(ctx, proj, voxelize_resolution=0.0); clipper.setFullScene(local_map)
    clipper.setRobotInLocalMap(guess); clipper.setSensorInRobot(S0)
    clipped = clipper.compute()
    al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.9), sensor_in_robot=S0, robustifier=api.RobustifierCauchy(0.01),
        min_num_correspondences=5, fixed_slice_name="points_0", moving_slice_name="points"))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=S1, min_num_correspondences=5,
        fixed_slice_name="points_1", moving_slice_name="points"))
    al.setFixed({"points_0": scans[0], "points_1": scans[1]}); al.setMoving({"points": clipped})
    al.setMovingInFixed([0, 0, 0])           # the clipped scene is already in the predicted robot frame
    assert al.compute() == 0
    # --- oracle pipeline
    oclip, _ = po.clip_scene(opr, m, guess, S0)
    assert np.array_equal(clipped.download(), oclip)
    osl = [_oracle_slice(po, s.slice_params()) for s in al.param_slice_processors]
    r = po.align(po.aligner_params(10), osl, scans, [oclip, oclip], np.zeros(3, np.float32))
    d = np.abs(al.movingInFixed() - r["pose"])
    assert r["status"] == 0 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    # corrected robot pose = guess * X^-1 ; it must be close to the true robot pose
    corrected = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(al.movingInFixed()[None, :].astype(np.float64)))[0]
    assert np.abs(corrected - robot[0])[:2].max() < 0.02 and abs(corrected[2] - robot[0][2]) < 0.01
    # --- merge the first scan at the corrected sensor pose, in place on the device
    sensor_in_map = synth.compose_poses(corrected[None, :], S0[None, :].astype(np.float64))[0].astype(np.float32)
    merger = api.MergerProjective2D(ctx, proj, 0.2); merger.setScene(local_map)
    merger.setMeasurement(scans[0]); merger.setMeasurementInScene(sensor_in_map)
    n = merger.compute()
    want, counts = po.merge_scene(opr, m, scans[0], sensor_in_map, 0.2)
    assert n == len(want) and merger.counts == counts and np.array_equal(local_map.download(), want)


def test_asynchronous_tracker_chain_equals_synchronous(ctx, po):
    """clip -> upload scans -> align -> merge x2 for several scans, once with every call synchronous and once with the clipper
    and the merger asynchronous (sizes known to the device only, one host synchronisation per step: the aligner's pose).
    Poses and the final local map must be IDENTICAL, and the map must equal the oracle's chain."""
    world = synth.make_world(3)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    S = [np.float32([0.2, 0.1, 0.1]), np.float32([-0.3, 0.0, math.pi])]
    robots = synth.sample_poses(world, 1, seed=5)
    traj = [robots[0]]
    for k in range(5):
        traj.append(synth.compose_poses(traj[-1][None, :], np.array([[0.05, 0.01, 0.02]]))[0])
    scans = [[synth.make_scans(world, synth.compose_poses(np.array([t]), s[None, :].astype(np.float64)), n_beams=721, noise_sigma=0.005, seed=17 + k)[0]
              for s in S] for k, t in enumerate(traj)]

    def run(asynchronous):
        local_map = api.CloudSet.reserved(ctx, 40000); local_map.upload(np.zeros((0, 4), np.float32))
        clipper = api.SceneClipperProjective2D(ctx, proj, asynchronous=asynchronous, voxelize_resolution=0.0); clipper.setFullScene(local_map)
        merger = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=asynchronous); merger.setScene(local_map)
        sets = [api.CloudSet.reserved(ctx, 1024), api.CloudSet.reserved(ctx, 1024)]
        al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
        for i, s in enumerate(S):
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
                api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=s, min_num_correspondences=5,
                fixed_slice_name="points_%d" % i, moving_slice_name="points"))
        est = traj[0].copy(); poses = []
        for i, s in enumerate(S):            # start: both scans merged at the true pose
            sets[i].upload(scans[0][i]); merger.setMeasurement(sets[i])
            merger.setMeasurementInScene(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0]); merger.compute()
        for k in range(1, len(traj)):
            guess = synth.compose_poses(est[None, :], np.array([[0.04, 0.0, 0.03]]))[0].astype(np.float32)
            clipper.setRobotInLocalMap(guess); clipper.setSensorInRobot(S[0])
            clipped = clipper.compute()
            for i in range(2):
                sets[i].upload(scans[k][i])
            al.setFixed({"points_0": sets[0], "points_1": sets[1]}); al.setMoving({"points": clipped}); al.setMovingInFixed([0, 0, 0])
            assert al.compute() == 0
            x = al.movingInFixed().astype(np.float64)
            est = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(x[None, :]))[0]
            poses.append(est.copy())
            for i, s in enumerate(S):
                merger.setMeasurement(sets[i]); merger.setMeasurementInScene(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0])
                merger.compute()
        return np.array(poses), local_map.download(), local_map.n_points

    p_sync, m_sync, n_sync = run(False)
    p_async, m_async, n_async = run(True)
    assert n_sync == n_async == len(m_sync) and n_sync > 400
    assert np.array_equal(p_sync, p_async) and np.array_equal(m_sync, m_async)
    assert np.abs(p_sync - np.array(traj[1:]))[:, :2].max() < 0.03
    # the same chain on the oracle (its poses feed its own merges; the GPU's differ by ~1e-7, so compare sizes and geometry)
    host_map = np.zeros((0, 4), np.float32); est = traj[0].copy()
    for i, s in enumerate(S):
        host_map, _ = po.merge_scene(opr, host_map, scans[0][i], np.float32(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0]), 0.2)
    osl = [po.slice_params(canvas_cols=721, range_max=20.0, normal_cos=0.8, min_num_correspondences=5, sensor_in_robot=tuple(s)) for s in S]
    for k in range(1, len(traj)):
        guess = synth.compose_poses(est[None, :], np.array([[0.04, 0.0, 0.03]]))[0].astype(np.float32)
        oclip, _ = po.clip_scene(opr, host_map, guess, S[0])
        r = po.align(po.aligner_params(10), osl, scans[k], [oclip, oclip], np.zeros(3, np.float32))
        est = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(r["pose"][None, :].astype(np.float64)))[0]
        assert np.abs(est - p_sync[k - 1])[:2].max() < 1e-4
        for i, s in enumerate(S):
            host_map, _ = po.merge_scene(opr, host_map, scans[k][i], np.float32(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0]), 0.2)
    assert abs(len(host_map) - n_sync) <= 0.02 * n_sync


def test_pending_sizes_are_resolved_where_the_host_needs_them(ctx, po):
    """Asynchronous clip / merge leave sizes on the device.  Every consumer must still be right: a point-query finder on a
    size-pending set (its grid needs the number), a download, a merge whose size BOUND no longer fits the capacity although the
    real size does, many asynchronous merges in a row (bound >> real size), and the host buffer of an upload reused at once."""
    world = synth.make_world(4)
    proj = api.PointNormal2fProjectorPolar(361, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(361, -math.pi, math.pi, 0.3, 20.0, 0.0)
    pose = synth.sample_poses(world, 1, seed=3)[0]
    scan = synth.make_scans(world, pose[None, :], n_beams=361, noise_sigma=0.004, seed=8)[0]
    # upload: the caller may overwrite its buffer right after the call
    buf = scan.copy()
    meas = api.CloudSet.reserved(ctx, 512); meas.upload(buf); buf[:] = 7.0
    assert np.array_equal(meas.download(), scan)
    # capacity just above what 40 merges of the SAME scan need (they mostly merge into existing points), far below 40 * cols
    local_map = api.CloudSet.reserved(ctx, 3 * 361); local_map.upload(np.zeros((0, 4), np.float32))
    merger = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=True); merger.setScene(local_map); merger.setMeasurement(meas)
    merger.setMeasurementInScene(pose.astype(np.float32))
    host_map = np.zeros((0, 4), np.float32)
    for _ in range(40):
        assert merger.compute() == -1
        host_map, _ = po.merge_scene(opr, host_map, scan, pose.astype(np.float32), 0.2)
    assert local_map.n_points == len(host_map) and np.array_equal(local_map.download(), host_map)
    # asynchronous clip, then consumers that need the exact size
    clipper = api.SceneClipperProjective2D(ctx, proj, asynchronous=True, voxelize_resolution=0.0); clipper.setFullScene(local_map)
    clipper.setRobotInLocalMap(pose.astype(np.float32)); clipper.setSensorInRobot([0, 0, 0])
    clipped = clipper.compute()
    oclip, _ = po.clip_scene(opr, host_map, pose.astype(np.float32), np.zeros(3, np.float32))
    kd = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.8)
    kd.setFixed(clipped); kd.setMoving(meas); kd.setLocalMapInSensor([0, 0, 0])         # fixed = the size-pending clipped set
    pairs = kd.compute()
    want = po.find(po.slice_params(finder=po.FINDER_NN, max_distance=0.3), oclip, scan, np.zeros(3, np.float32))
    assert np.array_equal(pairs, want) and len(pairs) > 100
    assert clipped.n_points == len(oclip) and np.array_equal(clipped.download(), oclip)
    # a second asynchronous clip straight from a size-pending scene (merge -> clip without any size query in between)
    merger.compute(); host_map, _ = po.merge_scene(opr, host_map, scan, pose.astype(np.float32), 0.2)
    clipped = clipper.compute()
    oclip, _ = po.clip_scene(opr, host_map, pose.astype(np.float32), np.zeros(3, np.float32))
    assert np.array_equal(clipped.download(), oclip)


# ---- RawDataPreprocessorProjective2D (row f2) --------------------------------------------------------------------
def test_preprocessor_reference_fixture_on_gpu(ctx):
    """tests/test_measurement_adaptor.cpp:10-39 on the device path: the Synthetic fixture gives exactly 100 points."""
    n = int(np.float32(1.0 - (-1.0)) / np.float32(0.02))
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.0, range_max=1000.0, voxelize_resolution=0.01)
    assert pre.setRawData(np.ones(n, np.float32), angle_min=-1.0, angle_max=1.0, range_min=0.0, range_max=1000.0)
    meas = pre.compute()
    assert meas.counts[0] == 100 and len(meas.download(0)) == 100
    # ... and the geometry the reference's own TODO asks for (tests/test_measurement_adaptor.cpp:38 "validate computed polar positions"):
    # every point on the unit circle, at the bearing of its beam (sensor matrix [[n / (angle_max - angle_min), n / 2]],
    # sensor_processing/raw_data_preprocessor_projective_2d.cpp:87-90), unit normals along the ray, facing the sensor
    pts = meas.download(0)
    assert np.allclose(np.hypot(pts[:, 0], pts[:, 1]), 1.0, atol=1e-6) and np.allclose(np.hypot(pts[:, 2], pts[:, 3]), 1.0, atol=1e-6)
    raw = api.RawDataPreprocessorProjective2D(ctx, range_min=0.0, range_max=1000.0, voxelize_resolution=0.0)
    raw.setRawData(np.ones(n, np.float32), angle_min=-1.0, angle_max=1.0, range_min=0.0, range_max=1000.0)
    rp = raw.compute().download(0)
    assert len(rp) == n and np.allclose(np.arctan2(rp[:, 1], rp[:, 0]), (np.arange(n) - n / 2) * (2.0 / n), atol=1e-6)
    dots = np.sum(rp[:, :2] * rp[:, 2:], 1)                            # a circle around the sensor: the normal is the (reversed) ray --
    assert np.all(dots < -0.98) and np.all(dots[20:-20] < -0.9999)     # exactly so away from the ends, where the sliding window is one-sided
    assert {tuple(np.round(p, 5)) for p in pts[:, :2]} == {tuple(np.round(p, 5)) for p in rp[:, :2]}     # 1 cm voxels keep all 100 (2 cm apart)


def test_preprocessor_bit_exact_batch_and_feeds_aligner(ctx, po):
    world = synth.make_world(2)
    poses = synth.sample_poses(world, 24, seed=4)
    a0, a1 = -2.34747, 2.35619                                       # laser_0 of MULTI.json:103-132 (asymmetric field of view)
    ranges = synth.make_scan_ranges(world, poses, n_beams=721, angle_min=a0, angle_max=a1, noise_sigma=0.005, seed=1)
    ranges[3, 100:140] = np.inf; ranges[5, :] = 0.01                  # a gap; a scan with every beam below range_min
    for vox, npd in ((0.02, 0.3), (0.0, 0.2), (0.1, 0.3)):
        pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=vox, normal_point_distance=npd)
        pre.setRawData(ranges, a0, a1, 0.0, 30.0)
        meas = pre.compute()
        pp = po.Preprocessor(721, a0, a1, 0.3, 20.0, npd, 5, vox)
        for i in range(len(poses)):
            want = po.preprocess_scan(pp, ranges[i])
            assert meas.counts[i] == len(want)
            assert np.array_equal(meas.download(i), want)
        assert meas.counts[5] == 0 and meas.counts.max() > 300
    # ranges in -> pose out, everything on the device: the preprocessed clouds are the aligner's fixed set
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    clean = synth.make_scan_ranges(world, poses, n_beams=721, angle_min=a0, angle_max=a1)
    pre.setRawData(clean, a0, a1, 0.0, 30.0)
    fixed = pre.compute()
    m = synth.make_map(world, 60000)
    x_true, x0 = synth.initial_guesses(poses, seed=9)
    al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(
        api.CorrespondenceFinderProjective2f(ctx, api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)), min_num_correspondences=10))
    res = al.compute_batch([fixed], [api.CloudSet(ctx, m)], x0.astype(np.float32))
    err = np.abs(res.pose - x_true)
    assert np.all(res.status == 0) and err[:, :2].max() < 3e-2 and err[:, 2].max() < 1e-2      # PCA normals on 2 cm voxels (corners!), not analytic ones
    # the same clouds through the oracle aligner give the same poses
    for i in (0, 7, 19):
        r = po.align(po.aligner_params(20), [po.slice_params(canvas_cols=721, range_max=20.0)], [fixed.download(i)], [m], x0[i].astype(np.float32))
        d = np.abs(res.pose[i] - r["pose"])
        assert d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD


def test_preprocessor_batch_form_and_single_scan_form_at_their_capacity_edges(ctx, po):
    """Round 5: a batch of >= 8 scans of <= 1152 beams runs the preprocessor's small form (512 threads, 34 KB: the sort's keys live where the unprojected points
    were, and pad to 2048 entries as soon as more than 1024 points carry a normal); anything else the one-beam-per-thread form.  Both against the oracle, bit for bit,
    on scans built to cross those edges: a smooth room seen with 1081 / 1150 / 1152 / 1153 / 2048 beams (every beam valid, nearly every point gets a normal: more than
    1024 of them), with and without voxelisation, as batches of 12 (small form where the beams fit) and of 3 (single-scan form)."""
    world = synth.make_world(4)
    poses = synth.sample_poses(world, 12, seed=11)
    for nb in (1081, 1150, 1152, 1153, 2048):
        a0, a1 = -0.75 * math.pi, 0.75 * math.pi
        ranges = synth.make_scan_ranges(world, poses, n_beams=nb, angle_min=a0, angle_max=a1, noise_sigma=0.002, seed=nb)
        for vox in (0.02, 0.0):
            pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.1, range_max=40.0, voxelize_resolution=vox, normal_point_distance=0.4)
            pp = po.Preprocessor(nb, a0, a1, 0.1, 40.0, 0.4, 5, vox)
            want = [po.preprocess_scan(pp, ranges[i]) for i in range(len(poses))]
            if vox == 0.0 and nb >= 1150:
                assert max(len(w) for w in want) > 1024      # (the case the key array must hold 2048 entries for)
            for lo, hi in ((0, 12), (3, 6)):
                pre.setRawData(ranges[lo:hi], a0, a1, 0.0, 50.0)
                meas = pre.compute()
                for i in range(lo, hi):
                    assert meas.counts[i - lo] == len(want[i]), (nb, vox, lo, i)
                    assert np.array_equal(meas.download(i - lo), want[i]), (nb, vox, lo, i)
                meas.close()


def test_preprocessor_reads_pinned_and_device_resident_ranges(ctx, po):
    """lsm2d_preprocess_scans takes its ranges from pageable host memory (staged), pinned host memory (copied from directly) or the
    device (read in place): the same clouds, bit for bit, and the oracle's."""
    import torch
    world = synth.make_world(2)
    poses = synth.sample_poses(world, 40, seed=14)
    a0, a1 = -2.34747, 2.35619
    ranges = synth.make_scan_ranges(world, poses, n_beams=1081, angle_min=a0, angle_max=a1, noise_sigma=0.005, seed=3)
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    got = []
    for src in (ranges, torch.from_numpy(ranges).pin_memory(), torch.from_numpy(ranges).to("cuda:0")):
        pre.setRawData(src, a0, a1, 0.0, 30.0)
        cs = pre.compute()
        got.append([cs.download(i) for i in range(len(poses))])
    pp = po.Preprocessor(1081, a0, a1, 0.3, 20.0, 0.3, 5, 0.02)
    for i in range(len(poses)):
        want = po.preprocess_scan(pp, ranges[i])
        assert len(want) > 200
        for g in got:
            assert np.array_equal(g[i], want), i
    with pytest.raises(ValueError):
        pre.setRawData(torch.from_numpy(ranges).to("cuda:0")[:1], a0, a1, 0.0, 30.0); pre.compute_into(api.CloudSet.reserved(ctx, 2048))


def test_ranges_in_pose_out_tracker_step_without_host_round_trips(ctx, po):
    """Row f2's point: raw ranges in, pose out, one synchronisation.  Two LaserMessages are preprocessed INTO reserved sets
    (lsm2d_preprocess_scan_into: same bits as the batched call), the local map is clipped, the aligner runs on the three
    size-pending sets, both measurements are merged -- every call but the aligner asynchronous.  Checked against the same
    chain on the oracle."""
    _ranges_in_pose_out_step(ctx, po)


def test_deferred_preprocessing_is_queued_by_the_first_reader(po):
    """Without kernel timing (the library's default) lsm2d_preprocess_scan_into only stages the ranges: the launch is queued by the
    set's first reader, and an aligner call that reads several such sets queues them as ONE launch (k_preprocess_multi, one workgroup
    per scan).  The same tracker step as above must come out bit for bit, and so must a set whose first reader is a size query, a
    download, a finder, or a second preprocessing call that replaces the first."""
    quiet = api.Context(0, kernel_timing=False)
    try:
        _ranges_in_pose_out_step(quiet, po)
        world = synth.make_world(6); a0, a1 = -2.34747, 2.35619
        rg = [synth.make_scan_ranges(world, synth.sample_poses(world, 1, seed=30 + i), n_beams=721, angle_min=a0, angle_max=a1, noise_sigma=0.004, seed=7 + i)[0] for i in range(3)]
        pp = po.Preprocessor(721, a0, a1, 0.3, 20.0, 0.3, 5, 0.02)
        want = [po.preprocess_scan(pp, r) for r in rg]
        pre = api.RawDataPreprocessorProjective2D(quiet, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
        st = api.CloudSet.reserved(quiet, 1024)
        pre.setRawData(rg[0], a0, a1, 0.0, 30.0); pre.compute_into(st)
        assert st.n_points == len(want[0])                                     # first reader: the size query
        pre.setRawData(rg[1], a0, a1, 0.0, 30.0); pre.compute_into(st)
        assert np.array_equal(st.download(), want[1])                          # first reader: the download
        pre.setRawData(rg[0], a0, a1, 0.0, 30.0); pre.compute_into(st)
        pre.setRawData(rg[2], a0, a1, 0.0, 30.0); pre.compute_into(st)         # replaces the scan nobody read
        m = synth.make_map(world, 5000, seed=2)
        f = api.CorrespondenceFinderProjective2f(quiet, api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0), 0.5, 0.8)
        f.setFixed(st); f.setMoving(m); f.setLocalMapInSensor(np.zeros(3, np.float32)); a = f.compute()      # first reader: the finder
        f.setFixed(want[2]); b = f.compute()
        assert np.array_equal(a, b) and np.array_equal(st.download(), want[2])
        st.upload(want[0]); assert np.array_equal(st.download(), want[0])       # an upload replaces a pending scan too
        pre.setRawData(rg[1], a0, a1, 0.0, 30.0); pre.compute_into(st); st.upload(want[2]); assert np.array_equal(st.download(), want[2])
    finally:
        quiet.close()


def test_clipper_and_merger_small_and_large_scene_paths(ctx, po):
    """Both implementations of the mapping steps -- one workgroup with LDS canvases (scenes <= 32768 points) and the
    many-workgroup split projection -- against the oracle, bit for bit."""
    world = synth.make_world(4)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    robot = synth.sample_poses(world, 1, seed=6)[0]
    S = np.float32([0.2, -0.1, 0.3])
    sensor = synth.compose_poses(robot[None, :], S[None, :].astype(np.float64))
    scan, _ = synth.make_scans(world, sensor, n_beams=721, noise_sigma=0.01, seed=2)
    for n_scene in (5000, 32768, 32769, 90000):
        m = synth.make_map(world, n_scene, noise_sigma=0.004, seed=n_scene)
        scene = api.CloudSet.reserved(ctx, n_scene + 2000); scene.upload(m)
        clipper = api.SceneClipperProjective2D(ctx, proj, voxelize_resolution=0.0); clipper.setFullScene(scene)
        clipper.setRobotInLocalMap(robot); clipper.setSensorInRobot(S)
        clipped = clipper.compute()
        want, wsrc = po.clip_scene(opr, m, np.float32(robot), S)
        assert np.array_equal(clipper.source_indices, wsrc) and np.array_equal(clipped.download(), want)
        merger = api.MergerProjective2D(ctx, proj, 0.2); merger.setScene(scene)
        merger.setMeasurement(scan); merger.setMeasurementInScene(np.float32(sensor[0]))
        n = merger.compute()
        wm, counts = po.merge_scene(opr, m, scan, np.float32(sensor[0]), 0.2)
        assert n == len(wm) and merger.counts == counts and np.array_equal(scene.download(), wm)


def test_randomised_mapping_and_preprocessing(ctx, po):
    """Fuzz the mapping side of the path over the parameters the ABI accepts -- projector geometry, sensor extrinsics, merge
    threshold, scene sizes on both sides of the one-workgroup limit, synchronous and asynchronous calls, preprocessor windows and
    voxel sizes -- bit for bit against the oracle.  LSM2D_FUZZ_TRIALS / LSM2D_FUZZ_SEED soak it."""
    import os
    n_trials = int(os.environ.get("LSM2D_FUZZ_TRIALS", "24")); rng = np.random.default_rng(int(os.environ.get("LSM2D_FUZZ_SEED", "77")))
    world = synth.make_world(8)
    maps = {n: synth.make_map(world, n, noise_sigma=0.004, seed=n + 1) for n in (700, 6000, 40000)}
    poses = synth.sample_poses(world, 10, seed=13)
    clipped_pts = merged_pts = prep_pts = 0
    for trial in range(n_trials):
        cols = int(rng.integers(90, 1500)); a0 = float(rng.uniform(-math.pi, -0.6)); a1 = float(rng.uniform(0.6, math.pi))
        rmin = float(rng.uniform(0.0, 0.8)); rmax = float(rng.uniform(6.0, 35.0)); off = float(rng.choice([0.0, 0.5]))
        S = np.float32([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-3, 3)]) if trial % 3 else np.zeros(3, np.float32)
        thr = float(rng.uniform(0.02, 0.5)); n_scene = (700, 6000, 40000)[trial % 3]; asynchronous = bool(trial % 2)
        robot = poses[trial % 10] + np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)])
        beams = int(rng.integers(64, 1400)); fov = float(rng.uniform(1.0, 3.1))
        vox = float(rng.choice([0.0, 0.02, 0.05, 0.2])); npd = float(rng.uniform(0.05, 0.6)); nmin = int(rng.integers(2, 9))
        m = maps[n_scene]
        proj = api.PointNormal2fProjectorPolar(cols, a0, a1, rmin, rmax, off); opr = po.Projector(cols, a0, a1, rmin, rmax, off)
        # preprocessor: raw ranges -> measurement (into a reserved set on odd trials)
        sensor = synth.compose_poses(robot[None, :], S[None, :].astype(np.float64))
        ranges = synth.make_scan_ranges(world, sensor, n_beams=beams, angle_min=-fov / 2, angle_max=fov / 2, noise_sigma=0.005, seed=trial)[0]
        pre = api.RawDataPreprocessorProjective2D(ctx, range_min=rmin, range_max=rmax, voxelize_resolution=vox, normal_point_distance=npd, normal_min_points=nmin)
        pre.setRawData(ranges, -fov / 2, fov / 2, 0.0, 40.0)
        meas_set = pre.compute_into(api.CloudSet.reserved(ctx, 2048)) if asynchronous else pre.compute()
        want_meas = po.preprocess_scan(po.Preprocessor(beams, -fov / 2, fov / 2, rmin, rmax, npd, nmin, vox), ranges)
        assert np.array_equal(meas_set.download(0), want_meas), ("preprocess", trial)
        prep_pts += len(want_meas)
        # clipper
        scene = api.CloudSet.reserved(ctx, n_scene + 4 * cols); scene.upload(m)
        clipper = api.SceneClipperProjective2D(ctx, proj, asynchronous=asynchronous, voxelize_resolution=0.0); clipper.setFullScene(scene)
        clipper.setRobotInLocalMap(np.float32(robot)); clipper.setSensorInRobot(S)
        clipped = clipper.compute()
        want_clip, want_src = po.clip_scene(opr, m, np.float32(robot), S)
        assert np.array_equal(clipped.download(), want_clip), ("clip", trial)
        if not asynchronous:
            assert np.array_equal(clipper.source_indices, want_src)
        clipped_pts += len(want_clip)
        # merger: the measurement twice (the second pass mostly merges into what the first one added)
        merger = api.MergerProjective2D(ctx, proj, thr, asynchronous=asynchronous); merger.setScene(scene); merger.setMeasurement(meas_set)
        mis = np.float32(sensor[0]); host = m
        for _ in range(2):
            merger.setMeasurementInScene(mis); n = merger.compute()
            host, counts = po.merge_scene(opr, host, want_meas, mis, thr)
            if not asynchronous:
                assert n == len(host) and merger.counts == counts, ("merge", trial)
        assert scene.n_points == len(host) and np.array_equal(scene.download(), host), ("merge", trial)
        merged_pts += len(host)
    print("mapping fuzz: %d trials, %d clipped / %d merged / %d preprocessed points bit-exact" % (n_trials, clipped_pts, merged_pts, prep_pts))
    assert clipped_pts > 1000 and prep_pts > 1000


def test_clipper_voxelize_branch_bit_exact(ctx, po):
    """SceneClipperProjective2D with voxelize_resolution > 0 (mapping/scene_clipper_projective_2d.cpp:36-48) on the device: bit-exact
    against the oracle for small and large scenes, several resolutions, with and without sensor extrinsics, synchronous and
    asynchronous; the voxelised scene then serves as the aligner's moving cloud."""
    world = synth.make_world(8)
    poses = synth.sample_poses(world, 4, seed=31)
    for n_scene, cols in ((700, 361), (6000, 721), (40000, 1081)):
        m = synth.make_map(world, n_scene, noise_sigma=0.004, seed=n_scene)
        proj = api.PointNormal2fProjectorPolar(cols, -math.pi, math.pi, 0.3, 20.0); opr = po.Projector(cols, -math.pi, math.pi, 0.3, 20.0, 0.0)
        scene = api.CloudSet.reserved(ctx, n_scene + 16); scene.upload(m)
        for k, res in enumerate((0.02, 0.05, 0.3)):
            robot = np.float32(poses[k]); S = np.float32([0.2, -0.1, 0.5]) if k % 2 else np.zeros(3, np.float32)
            for asynchronous in (False, True):
                clipper = api.SceneClipperProjective2D(ctx, proj, voxelize_resolution=res, asynchronous=asynchronous)
                clipper.setFullScene(scene); clipper.setRobotInLocalMap(robot); clipper.setSensorInRobot(S)
                got = clipper.compute().download()
                want = po.clip_scene_voxelized(opr, m, robot, S, res)
                assert len(want) > 10 and len(got) == len(want) and np.array_equal(got, want), (n_scene, res, asynchronous)
    with pytest.raises(api.Lsm2dError):           # voxelisation is limited to 2048 columns
        c = api.SceneClipperProjective2D(ctx, api.PointNormal2fProjectorPolar(4096, -math.pi, math.pi, 0.3, 20.0), voxelize_resolution=0.1)
        c.setFullScene(scene); c.compute()


def test_merge_into_large_scene_with_pending_measurement_count(ctx, po):
    """A measurement whose size only the device knows (lsm2d_preprocess_scan_into, no download) merged into a scene beyond the
    one-workgroup limit: the multi-launch merge path takes sizes by value, so the pending count must be resolved first -- an upper
    bound (n_beams) would push the stale tail a LONGER earlier scan left in the same reserved set through the merge.  And the mirror
    case: a scene whose size is pending (asynchronous merge before) with a measurement beyond the limit."""
    world = synth.make_world(8)
    m = synth.make_map(world, 40000, noise_sigma=0.004, seed=3)
    poses = synth.sample_poses(world, 2, seed=21)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0); opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    beams, fov = 1081, 2.3
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.0, normal_point_distance=0.3, normal_min_points=5)
    opre = po.Preprocessor(beams, -fov / 2, fov / 2, 0.3, 20.0, 0.3, 5, 0.0)
    meas_set = api.CloudSet.reserved(ctx, 2048)
    # first a full scan fills the set's slots ...
    r_long = synth.make_scan_ranges(world, poses[:1], n_beams=beams, angle_min=-fov / 2, angle_max=fov / 2, seed=1)[0]
    pre.setRawData(r_long, -fov / 2, fov / 2, 0.0, 40.0); pre.compute_into(meas_set)
    assert len(meas_set.download(0)) > 900
    # ... then a scan with two thirds of its beams out of range reuses it: real count ~1/3, upper bound still n_beams, no download
    r_short = synth.make_scan_ranges(world, poses[1:2], n_beams=beams, angle_min=-fov / 2, angle_max=fov / 2, seed=2)[0].copy()
    r_short[: 2 * beams // 3] = np.inf
    pre.setRawData(r_short, -fov / 2, fov / 2, 0.0, 40.0); pre.compute_into(meas_set)
    want_meas = po.preprocess_scan(opre, r_short)
    assert 0 < len(want_meas) < 500
    scene = api.CloudSet.reserved(ctx, len(m) + 4 * 721); scene.upload(m)
    merger = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=True); merger.setScene(scene); merger.setMeasurement(meas_set)
    mis = np.float32(poses[1]); merger.setMeasurementInScene(mis); merger.compute()
    host, _ = po.merge_scene(opr, m, want_meas, mis, 0.2)
    got = scene.download()
    assert len(got) == len(host) and np.array_equal(got, host)
    # mirror case: pending scene size (<= 32768 bound) and an exact measurement beyond the limit
    small = synth.make_map(world, 6000, noise_sigma=0.004, seed=5)
    big_meas_pts = synth.make_map(world, 36000, noise_sigma=0.004, seed=6)
    scene2 = api.CloudSet.reserved(ctx, 6000 + 40 * 721); scene2.upload(small)
    merger2 = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=True); merger2.setScene(scene2); merger2.setMeasurement(meas_set)
    merger2.setMeasurementInScene(mis); merger2.compute()                  # leaves scene2's size pending
    host2, _ = po.merge_scene(opr, small, want_meas, mis, 0.2)
    big = api.CloudSet(ctx, big_meas_pts)
    ident = np.zeros(3, np.float32)
    merger3 = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=True); merger3.setScene(scene2); merger3.setMeasurement(big)
    merger3.setMeasurementInScene(ident); merger3.compute()
    host3, _ = po.merge_scene(opr, host2, big_meas_pts, ident, 0.2)
    got3 = scene2.download()
    assert len(got3) == len(host3) and np.array_equal(got3, host3)
