"""CPU tests of the oracle (the restated reference algorithm): known answers and invariants.

The reference has no golden vector for this path -- these are the substitutes it
lists: hand-checkable projector cases, finite-difference Jacobian, the SE(2) restriction of
octave/solver/nicp_post.m evaluated independently (tests/golden/nicp_2d_known_answer.json), zero-noise
convergence to the generating pose, the status logic of the aligner."""
import json
import math

import numpy as np
import pytest

from conftest import golden_path
from srrg2_laser_slam_2d_amd import synth


def test_atan2_polynomial_accuracy(po):
    rng = np.random.default_rng(0)
    y = rng.normal(size=20000).astype(np.float32); x = rng.normal(size=20000).astype(np.float32)
    got = po.atan2f(y, x)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    assert np.max(np.abs(got - ref)) < 3.5e-7
    # axes, origin, signs
    for (yy, xx, want) in [(0, 0, 0.0), (0, 1, 0.0), (1, 0, math.pi / 2), (0, -1, math.pi), (-1, 0, -math.pi / 2),
                           (1, 1, math.pi / 4), (-1, -1, -3 * math.pi / 4), (1e-30, 1e30, 0.0)]:
        assert abs(float(po.atan2f([yy], [xx])[0]) - want) < 3e-7


def test_bearing_follows_the_device_quotient_on_the_four_known_ties(po):
    """The oracle's bearing divides min(|x|,|y|) by r = sqrtf(x^2 + y^2) -- IEEE, except where the device's fused quotient sequence is
    provably one ulp low (r with an all-ones mantissa, numerator a power of two: tools/fp_exact_check.hip, profiles/r02/fp_exact_full_r02e.log)."""
    import struct
    f32 = lambda v: struct.unpack("f", struct.pack("f", v))[0]
    r_all_ones = struct.unpack("f", struct.pack("I", 0x3FFFFFFF))[0]          # 2 - 2^-23
    y = 0.5; x = math.sqrt(r_all_ones * r_all_ones - y * y)
    # search the float x for which fmaf(x, x, y*y) rounds to an r2 whose square root is the all-ones r
    xs = np.nextafter(np.float32(x), np.float32([0, 4]))
    hit = False
    for xv in [np.float32(x), xs[0], xs[1]]:
        r2 = np.float32(np.float64(xv) * np.float64(xv) + np.float64(np.float32(y)) * np.float64(np.float32(y)))
        if struct.unpack("I", struct.pack("f", float(np.sqrt(r2, dtype=np.float32))))[0] == 0x3FFFFFFF:
            got = float(po.atan2f([y], [float(xv)])[0])
            t_ieee = np.float32(y) / np.sqrt(r2, dtype=np.float32)
            t_dev = np.nextafter(t_ieee, np.float32(0))
            # the bearing is asin(t): the two candidates differ by ~3e-8 rad; the oracle must sit on the device's side
            assert abs(got - math.asin(float(t_dev))) <= abs(got - math.asin(float(t_ieee))) + 1e-9
            hit = True
    assert hit


def _pt(r, ang, nx=1.0, ny=0.0):
    return [r * math.cos(ang), r * math.sin(ang), nx, ny]


def test_projector_columns_ties_and_gates(po):
    pr = po.Projector(360, -math.pi, math.pi, 0.5, 10.0, 0.0)      # 1 degree per column, column 180 = angle 0
    deg = math.pi / 180
    cloud = np.array([
        _pt(2.0, 0.5 * deg),      # 0 -> column 180
        _pt(1.5, 0.6 * deg),      # 1 -> column 180, nearer: wins
        _pt(1.5, 0.7 * deg),      # 2 -> column 180, same... different float depth, see below
        _pt(0.4, 10.5 * deg),     # 3 below range_min: dropped
        _pt(11.0, 20.5 * deg),    # 4 above range_max: dropped
        _pt(3.0, -90.5 * deg),    # 5 -> column 89
        _pt(3.0, 179.5 * deg),    # 6 -> column 359
        _pt(3.0, -179.5 * deg),   # 7 -> column 0
    ], np.float32)
    src, depth, xyn = po.project(pr, cloud, [0, 0, 0])
    assert src[180] in (1, 2) and abs(depth[180] - 1.5) < 1e-6
    assert src[190] == -1 and src[200] == -1
    assert src[89] == 5 and src[359] == 6 and src[0] == 7
    assert (src >= 0).sum() == 4
    assert depth[5] == np.finfo(np.float32).max
    # exact tie: identical points -> the first index wins (strict <)
    tie = np.array([_pt(2.0, 30.5 * deg)] * 3, np.float32)
    s2, _, _ = po.project(pr, tie, [0, 0, 0])
    assert s2[210] == 0
    # the pose maps points into the camera frame: rotate by +90 deg moves angle 0.5deg to 90.5deg
    s3, d3, x3 = po.project(pr, cloud[:1], [0, 0, math.pi / 2])
    assert s3[270] == 0 and abs(x3[270, 3] - 1.0) < 1e-6     # normal (1,0) -> (0,1)
    # col_offset = 0.5 rounds to nearest
    pr2 = po.Projector(360, -math.pi, math.pi, 0.5, 10.0, 0.5)
    s4, _, _ = po.project(pr2, cloud[:1], [0, 0, 0])
    assert s4[181] == 0


def test_projective_finder_identity_and_gates(po):
    deg = math.pi / 180
    ang = (np.arange(-60, 60) + 0.5) * deg
    fixed = np.stack([3 * np.cos(ang), 3 * np.sin(ang), -np.cos(ang), -np.sin(ang)], 1).astype(np.float32)
    sp = po.slice_params(canvas_cols=360, range_min=0.3, range_max=20, point_distance=0.5, normal_cos=0.8)
    c = po.find(sp, fixed, fixed, [0, 0, 0])
    assert len(c) == 120 and np.array_equal(c[:, 0], c[:, 1]) and np.all(np.diff(c[:, 0]) > 0)   # ascending column
    # depth gate: moving pushed 0.6 m away radially -> no pairs
    far = fixed.copy(); far[:, :2] *= 3.6 / 3.0
    assert len(po.find(sp, fixed, far, [0, 0, 0])) == 0
    # normal gate: moving normals rotated by 45 deg (cos = 0.707 < 0.8)
    rot = fixed.copy(); c45 = math.cos(math.pi / 4)
    rot[:, 2] = c45 * fixed[:, 2] - c45 * fixed[:, 3]; rot[:, 3] = c45 * fixed[:, 2] + c45 * fixed[:, 3]
    assert len(po.find(sp, fixed, rot, [0, 0, 0])) == 0
    # empty clouds
    assert len(po.find(sp, fixed[:0], fixed, [0, 0, 0])) == 0
    assert len(po.find(sp, fixed, fixed[:0], [0, 0, 0])) == 0


def test_nn_finder_grid_equals_brute_force(po):
    rng = np.random.default_rng(1)
    def cloud(n):
        p = rng.uniform(-5, 5, size=(n, 2)); a = rng.uniform(-np.pi, np.pi, n)
        return np.concatenate([p, np.cos(a)[:, None], np.sin(a)[:, None]], 1).astype(np.float32)
    fixed, moving = cloud(3000), cloud(2000)
    fixed[10] = fixed[11]                       # duplicate point: tie -> lowest index
    moving[0, :2] = fixed[11, :2]
    for md in (0.05, 0.3, 1.0):
        sp = po.slice_params(finder=po.FINDER_NN, max_distance=md, normal_cos=-1.0)
        g = po.find(sp, fixed, moving, [0.1, -0.2, 0.3]); b = po.find(sp, fixed, moving, [0.1, -0.2, 0.3], brute=True)
        assert np.array_equal(g, b) and len(g) > 0
    sp = po.slice_params(finder=po.FINDER_NN, max_distance=0.3, normal_cos=-1.0)
    c = po.find(sp, fixed, moving, [0, 0, 0])
    assert c[0, 1] == 0 and c[0, 0] == 10
    # ascending moving index, normal gate prunes
    assert np.all(np.diff(c[:, 1]) > 0)
    sp2 = po.slice_params(finder=po.FINDER_NN, max_distance=0.3, normal_cos=0.8)
    assert 0 < len(po.find(sp2, fixed, moving, [0, 0, 0])) < len(c)


def test_factor_jacobian_finite_differences(po):
    rng = np.random.default_rng(2)
    for _ in range(20):
        f = rng.normal(size=4); m = rng.normal(size=4)
        f[2:] /= np.linalg.norm(f[2:]); m[2:] /= np.linalg.norm(m[2:])
        f = f.astype(np.float32); m = m.astype(np.float32)
        pose = rng.uniform(-1, 1, 3)
        e, J = po.error_jacobian(f, m, pose)
        for k in range(3):
            d = np.zeros(3); d[k] = 1e-6
            def at(dx):
                c, s = math.cos(pose[2]), math.sin(pose[2])
                p = [pose[0] + c * dx[0] - s * dx[1], pose[1] + s * dx[0] + c * dx[1], pose[2] + dx[2]]   # X * v2t(dx)
                return po.error_jacobian(f, m, p)[0]
            num = (at(d) - at(-d)) / 2e-6
            assert np.allclose(num, J[:, k], atol=1e-7)


def test_known_answer_nicp_restriction(po):
    g = json.load(open(golden_path("nicp_2d_known_answer.json")))
    fixed = np.array(g["fixed"], np.float32); moving = np.array(g["moving"], np.float32)
    corr = np.array([[0, 0], [1, 1], [2, 2]], np.int32)
    sp = po.slice_params()
    for i in range(3):
        e, J = po.error_jacobian(fixed[i], moving[i], g["pose"])
        assert np.allclose(e, g["pairs"][i]["e"], atol=1e-7) and np.allclose(J, g["pairs"][i]["J"], atol=1e-7)
    H, b, st = po.linearize(sp, fixed, moving, corr, g["pose"], double=True)
    assert np.allclose(H, g["H"], atol=1e-6) and np.allclose(b, g["b"], atol=1e-6)
    assert st.n_corr == 3 and st.n_in == 3 and abs(st.chi_in - g["chi"]) < 1e-6
    Hf, bf, _ = po.linearize(sp, fixed, moving, corr, g["pose"], double=False)
    assert np.allclose(Hf, g["H"], atol=1e-5) and np.allclose(bf, g["b"], atol=1e-5)
    rc, pose, dx = po.solve_update(H, b, g["pose"], double=True)
    assert rc == 0 and np.allclose(dx, g["dx"], atol=1e-5) and np.allclose(pose, g["pose_after_step"], atol=1e-5)
    # Cauchy robustifier
    spc = po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=g["cauchy"]["tau"])
    Hc, bc, stc = po.linearize(spc, fixed, moving, corr, g["pose"], double=True)
    assert np.allclose(Hc, g["cauchy"]["H"], atol=1e-6) and np.allclose(bc, g["cauchy"]["b"], atol=1e-6)
    assert stc.n_in == g["cauchy"]["n_inliers"] and stc.n_out == 3 - g["cauchy"]["n_inliers"]
    assert abs(stc.chi_in - g["cauchy"]["chi_inliers"]) < 1e-6 and abs(stc.chi_out - g["cauchy"]["chi_outliers"]) < 1e-6


def test_h22_b2_identity(po):
    """the normal rows only touch H22 (+= w|n_m|^2) and b2 (= -w (R J2 n_m).n_f)."""
    f = np.array([[1, 2, 0.6, 0.8]], np.float32); m = np.array([[1.1, 1.9, 0.8, 0.6]], np.float32)
    pose = [0.1, 0.2, 0.3]
    e, J = po.error_jacobian(f[0], m[0], pose)
    c, s = math.cos(0.3), math.sin(0.3)
    nq = np.array([c * 0.8 - s * 0.6, s * 0.8 + c * 0.6]); d = np.array([-nq[1], nq[0]])
    assert np.allclose(J[1:, 2], d, atol=1e-6) and np.allclose(J[1:, :2], 0)
    assert abs(d @ e[1:] - (-(d @ f[0, 2:]))) < 1e-6


def test_solve_singular_and_damping(po):
    H = np.diag([0.0, 1.0, 1.0]); b = np.ones(3)
    rc, _, _ = po.solve_update(H, b, [0, 0, 0], double=True)
    assert rc == po.SINGULAR_H
    rc, pose, dx = po.solve_update(H, b, [0, 0, 0], damping=1.0, double=True)
    assert rc == 0 and np.allclose(dx, [-1, -0.5, -0.5])
    # right update: translation is rotated by the current heading; angle wraps into (-pi, pi]
    rc, pose, dx = po.solve_update(np.eye(3), [-1.0, 0.0, -0.2], [0, 0, 3.1], double=True)
    assert abs(pose[0] - math.cos(3.1)) < 1e-12 and abs(pose[1] - math.sin(3.1)) < 1e-12 and abs(pose[2] - (3.3 - 2 * math.pi)) < 1e-12


@pytest.mark.parametrize("double", [False, True])
def test_aligner_converges_to_generating_pose(po, small_workload, double):
    wl = small_workload
    sp = po.slice_params(); ap = po.aligner_params(20)
    for i in range(4):
        f = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        r = po.align(ap, [sp], [f], [wl.map_points], wl.x0[i], double=double)
        assert r["status"] == po.SUCCESS and r["iterations"] == 20
        err = np.abs(r["pose"] - wl.x_true[i])
        assert err[:2].max() < (2e-6 if double else 2e-5) and err[2] < (1e-6 if double else 1e-5)
        assert r["stats"][-1].chi_in < 1e-6 and r["stats"][-1].n_corr > 300
        assert np.allclose(r["H"], r["H"].T) and np.all(np.linalg.eigvalsh(r["H"]) > 0)


def test_aligner_nn_finder_converges(po, small_workload):
    wl = small_workload
    sp = po.slice_params(finder=po.FINDER_NN, max_distance=0.5)
    ap = po.aligner_params(20)
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    # role B: fixed = map, moving = scan; the estimate is then scan-in-map = inverse of x_true
    x_true_b = synth.invert_poses(wl.x_true[:1])[0]; x0_b = synth.invert_poses(wl.x0[:1].astype(np.float64))[0]
    r = po.align(ap, [sp], [wl.map_points], [f], x0_b, double=True)
    assert r["status"] == po.SUCCESS
    assert np.abs(r["pose"] - x_true_b)[:2].max() < 5e-3 and abs(r["pose"][2] - x_true_b[2]) < 2e-3


def test_aligner_status_logic(po, small_workload):
    wl = small_workload
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    sp = po.slice_params(); ap = po.aligner_params(20)
    # far-off initial guess: nothing matches -> NotEnoughCorrespondences on the first iteration, pose untouched
    r = po.align(ap, [sp], [f], [wl.map_points + np.float32([100, 100, 0, 0])], wl.x0[0])
    assert r["status"] == po.NOT_ENOUGH_CORRESPONDENCES and r["iterations"] == 1 and np.allclose(r["pose"], wl.x0[0])
    # min_num_inliers above what any scan can give
    r = po.align(po.aligner_params(20, min_num_inliers=100000), [sp], [f], [wl.map_points], wl.x0[0])
    assert r["status"] == po.NOT_ENOUGH_INLIERS
    # a single wall: translation along it is unobservable -> H singular at heading 0
    wall = np.stack([np.linspace(-3, 3, 400), np.full(400, 2.0), np.zeros(400), -np.ones(400)], 1).astype(np.float32)
    r = po.align(ap, [po.slice_params(min_num_correspondences=0)], [wall], [wall], [0, 0, 0], double=True)
    assert r["status"] == po.SINGULAR_H
    # zero iterations: Success, pose = initial guess
    r = po.align(po.aligner_params(0), [sp], [f], [wl.map_points], wl.x0[0])
    assert r["status"] == po.SUCCESS and r["iterations"] == 0 and np.allclose(r["pose"], wl.x0[0])


def test_cauchy_reduces_outlier_influence(po, small_workload):
    wl = small_workload
    f = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]].copy()
    f[::7, :2] += 0.3 * f[::7, 2:]         # every 7th scan point displaced 30 cm along its normal
    ap = po.aligner_params(20)
    plain = po.align(ap, [po.slice_params()], [f], [wl.map_points], wl.x0[1], double=True)
    cauchy = po.align(ap, [po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=0.05)], [f], [wl.map_points], wl.x0[1], double=True)
    e_plain = np.abs(plain["pose"] - wl.x_true[1])[:2].max(); e_cauchy = np.abs(cauchy["pose"] - wl.x_true[1])[:2].max()
    assert e_cauchy < e_plain and cauchy["stats"][-1].n_out > 50


def test_multi_slice_with_sensor_offsets_and_prior(po):
    """MULTI.json:715-721: two laser slices with their own extrinsics (+ an odometry prior) share ONE pose."""
    world = synth.make_world(5)
    m = synth.make_map(world, 30000)
    robot = synth.sample_poses(world, 1, seed=11)
    S0, S1 = np.array([0.2, 0.1, 0.1]), np.array([-0.3, 0.0, math.pi])
    scans = []
    for S in (S0, S1):
        sensor = synth.compose_poses(robot, S[None, :])
        pts, offs = synth.make_scans(world, sensor, n_beams=721)
        scans.append(pts)
    x_true = synth.invert_poses(robot)[0]                      # map in robot frame
    delta = np.array([[0.04, -0.03, 0.03]])
    x0 = synth.invert_poses(synth.compose_poses(robot, delta))[0]
    sl = [po.slice_params(canvas_cols=721, range_max=20, sensor_in_robot=tuple(S), min_num_correspondences=5,
                          robustifier=po.ROBUST_CAUCHY if k == 0 else po.ROBUST_NONE, chi_threshold=0.01)
          for k, S in enumerate((S0, S1))]
    ap = po.aligner_params(10)
    r = po.align(ap, sl, scans, [m, m], x0, double=True)
    assert r["status"] == po.SUCCESS
    assert np.abs(r["pose"] - x_true)[:2].max() < 1e-4 and abs(r["pose"][2] - x_true[2]) < 1e-4
    assert r["stats"][-1].n_corr > 600        # both slices contribute
    # a stiff prior at the initial guess holds the estimate there
    app = po.aligner_params(10, prior_z=x0, prior_omega=np.eye(3) * 1e9)
    rp = po.align(app, sl, scans, [m, m], x0, double=True)
    assert np.abs(rp["pose"] - x0).max() < 1e-4
    # one slice starved of correspondences is skipped, the other still aligns
    sl2 = [sl[0], po.slice_params(canvas_cols=721, range_max=20, sensor_in_robot=tuple(S1), min_num_correspondences=10000)]
    r2 = po.align(ap, sl2, scans, [m, m], x0, double=True)
    assert r2["status"] == po.SUCCESS and np.abs(r2["pose"] - x_true)[:2].max() < 1e-3


def test_batch_driver_matches_single_calls(po, small_workload):
    wl = small_workload
    sp = po.slice_params(); ap = po.aligner_params(5)
    xo, H, status, last = po.align_batch(ap, sp, wl.scan_points, wl.scan_offsets, wl.map_points, wl.x0, n_threads=3)
    for i in range(len(wl.x0)):
        f = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        r = po.align(ap, [sp], [f], [wl.map_points], wl.x0[i])
        assert np.array_equal(r["pose"], xo[i]) and status[i] == r["status"] and last[i].n_corr == r["stats"][-1].n_corr


def test_clipper_and_merger_semantics(po):
    """mapping/scene_clipper_projective_2d.cpp:11-65 and mapping/merger_projective_2d.cpp:9-100 restated."""
    pr = po.Projector(360, -math.pi, math.pi, 0.3, 20.0, 0.0)
    deg = math.pi / 180
    def pt(r, a, nx=-1.0, ny=0.0):
        return [r * math.cos(a), r * math.sin(a), nx, ny]
    scene = np.array([pt(5, 0.5 * deg), pt(7, 0.6 * deg), pt(5, 10.5 * deg), pt(25, 20.5 * deg), pt(5, 30.5 * deg)], np.float32)
    clipped, src = po.clip_scene(pr, scene, [0, 0, 0])
    assert list(src) == [0, 2, 4]                          # occluded point 1 and out-of-range point 3 are dropped
    assert np.allclose(clipped, scene[[0, 2, 4]], atol=1e-6)
    # robot rotated by 90 deg + sensor offset: clipped points come back in the ROBOT frame
    robot = np.array([1.0, -2.0, math.pi / 2]); S = np.array([0.3, 0.1, 0.2])
    clipped, src = po.clip_scene(pr, scene, robot, S, double=True)
    c, s = math.cos(robot[2]), math.sin(robot[2])
    back = np.stack([robot[0] + c * clipped[:, 0] - s * clipped[:, 1], robot[1] + s * clipped[:, 0] + c * clipped[:, 1]], 1)
    assert np.allclose(back, scene[src, :2], atol=1e-5)
    # merger: measurement seen from the scene origin
    meas = np.array([pt(5.1, 0.5 * deg),      # within 0.2 of scene depth 5 -> merged
                     pt(6.0, 10.5 * deg),     # behind the scene point (dr = +1) -> replaces it
                     pt(3.0, 30.5 * deg),     # in front (dr = -2) -> appended
                     pt(4.0, 50.5 * deg),     # empty column -> appended as new
                     pt(19.0, 60.5 * deg)],   # deeper than 0.9 * range_max -> ignored
                    np.float32)
    new_scene, counts = po.merge_scene(pr, scene, meas, [0, 0, 0], merge_threshold=0.2)
    assert counts == (1, 1, 1) and len(new_scene) == len(scene) + 2
    assert np.allclose(new_scene[0, :2], (scene[0, :2] + meas[0, :2]) / 2, atol=1e-6) and abs(np.linalg.norm(new_scene[0, 2:]) - 1) < 1e-6
    assert np.allclose(new_scene[2], meas[1], atol=1e-6)
    assert np.allclose(new_scene[5], meas[2], atol=1e-6) and np.allclose(new_scene[6], meas[3], atol=1e-6)   # appended in ascending column
    assert np.array_equal(new_scene[[1, 3, 4]], scene[[1, 3, 4]])


def test_clipper_voxelize_branch_semantics(po):
    """mapping/scene_clipper_projective_2d.cpp:36-48: with voxelize_resolution > 0 the clipped points are voxelised in the SENSOR frame
    with coefficients (res, res, 0.1, 0.1) and then moved to the robot frame; resolution <= 0 is the plain clipper."""
    wl = synth.make_workload(1, 6000, seed=4)
    pr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    robot = np.float32([0.4, -0.3, 0.2]); S = np.float32([0.15, 0.05, -0.4])
    plain, _ = po.clip_scene(pr, wl.map_points, robot, S)
    assert np.array_equal(po.clip_scene_voxelized(pr, wl.map_points, robot, S, 0.0), plain)
    fine = po.clip_scene_voxelized(pr, wl.map_points, robot, S, 2e-3)        # voxels far smaller than the point spacing: a re-ordering only
    assert len(fine) == len(plain)
    assert np.allclose(np.sort(fine[:, 0]), np.sort(plain[:, 0]), atol=2e-6) and np.allclose(np.sort(fine[:, 1]), np.sort(plain[:, 1]), atol=2e-6)
    coarse = po.clip_scene_voxelized(pr, wl.map_points, robot, S, 0.25)
    assert 20 < len(coarse) < 0.7 * len(plain)
    assert np.allclose(np.hypot(coarse[:, 2], coarse[:, 3]), 1.0, atol=1e-5)
    # voxel keys are formed in the sensor frame: undo S and check one point per (x, y, nx, ny) voxel, ascending key order
    Si = synth.invert_poses(S[None, :].astype(np.float64))[0]
    c, s_ = math.cos(Si[2]), math.sin(Si[2])
    xs = c * coarse[:, 0] - s_ * coarse[:, 1] + Si[0]; ys = s_ * coarse[:, 0] + c * coarse[:, 1] + Si[1]
    kx = np.floor(xs / 0.25); ky = np.floor(ys / 0.25)
    order = kx * 1e6 + ky
    assert np.all(np.diff(order) >= 0)                        # ascending (x, y) voxel; ties differ in the normal's voxel
    # every clipped point falls into the voxel of exactly one output point's (x, y) cell or a neighbour (averaging stays inside a voxel)
    plain_s, _ = po.clip_scene(pr, wl.map_points, np.float32(synth.compose_poses(robot[None, :].astype(np.float64), S[None, :].astype(np.float64))[0]), np.zeros(3, np.float32))
    cells = {(int(np.floor(p[0] / 0.25)), int(np.floor(p[1] / 0.25))) for p in plain_s}
    assert {(int(a), int(b)) for a, b in zip(kx, ky)} <= cells


def test_preprocessor_reference_fixture_and_semantics(po):
    """Row f2.  The one value the reference's own tests pin on this path: the `Synthetic` fixture (tests/fixtures.hpp:8-53:
    (1 - -1)/0.02 beams, every range 1 m, range limits [0, 1000], voxelize 0.01) must give exactly 100 points
    (tests/test_measurement_adaptor.cpp:36)."""
    n = int(np.float32(1.0 - (-1.0)) / np.float32(0.02))
    pp = po.Preprocessor(n, -1.0, 1.0, 0.0, 1000.0, 0.3, 5, 0.01)
    pts = po.preprocess_scan(pp, np.ones(n, np.float32))
    assert len(pts) == 100
    assert np.allclose(np.hypot(pts[:, 0], pts[:, 1]), 1.0, atol=1e-6)                  # on the unit circle
    assert np.all(np.sum(pts[:, :2] * pts[:, 2:], 1) < -0.98)                            # normals face the sensor
    assert np.allclose(np.hypot(pts[:, 2], pts[:, 3]), 1.0, atol=1e-6)
    # bearings follow the sensor matrix [[1/res, n/2]]: beam c looks along (c - n/2) * res
    raw = po.preprocess_scan(po.Preprocessor(n, -1.0, 1.0, 0.0, 1000.0, 0.3, 5, 0.0), np.ones(n, np.float32))
    ang = np.arctan2(raw[:, 1], raw[:, 0])
    assert np.allclose(ang, (np.arange(n) - n / 2) * (2.0 / n), atol=1e-6)
    # range gates, too few neighbours, voxel merging
    r = np.ones(n, np.float32); r[10] = 2000.0; r[20:23] = -1.0
    assert len(po.preprocess_scan(po.Preprocessor(n, -1.0, 1.0, 0.0, 1000.0, 0.3, 5, 0.0), r)) == n - 4
    assert len(po.preprocess_scan(po.Preprocessor(n, -1.0, 1.0, 0.0, 1000.0, 0.01, 5, 0.0), np.ones(n, np.float32))) == 0     # 2 cm spacing, 1 cm window
    coarse = po.preprocess_scan(po.Preprocessor(n, -1.0, 1.0, 0.0, 1000.0, 0.3, 5, 0.1), np.ones(n, np.float32))
    assert 15 < len(coarse) < 40
    key = np.floor(coarse[:, 0] / np.float32(0.1)) * 1e6 + np.floor(coarse[:, 1] / np.float32(0.1))
    assert np.all(np.diff(key) > 0)                                                      # ascending voxel order, one point per voxel


def test_oracle_regression_vectors(po):
    """tests/golden/oracle_regression.json freezes the restated algorithm on small seeded inputs (oracle-generated, NOT
    reference outputs).  The fp32 mirror is a fixed sequence of IEEE operations (no libm call): its poses, counts and chi^2 sums are
    frozen BIT FOR BIT, in both summation orders; the fp64 run goes through the host's libm and is held to 1e-6."""
    g = json.load(open(golden_path("oracle_regression.json")))
    wl = synth.make_workload(3, 8000, seed=42, n_beams=361)
    sps = {"projective": po.slice_params(canvas_cols=361), "nn": po.slice_params(finder=po.FINDER_NN, max_distance=0.3),
           "distmap": po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.1)}
    for c in g["cases"]:
        i = c["index"]; scan = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        for name, sp in sps.items():
            want = c[name]
            pairs = po.find(sp, scan, wl.map_points, wl.x0[i])
            assert len(pairs) == want["n_pairs"]
            assert int((pairs.astype(np.int64) * np.array([1000003, 7919])).sum() % (2 ** 31)) == want["pairs_checksum"]
            for tag, dev in (("sequential", False), ("device_order", True)):
                w32 = c[name + "_fp32"][tag]
                rf = po.align(po.aligner_params(10, device_order=dev), [sp], [scan], [wl.map_points], wl.x0[i])
                assert rf["status"] == w32["status"] and [float(v).hex() for v in rf["pose"]] == w32["pose_hex"], (name, tag, i)
                assert [int(st.n_corr) for st in rf["stats"]] == w32["n_corr"] and [float(st.chi_in).hex() for st in rf["stats"]] == w32["chi_in_hex"]
            r = po.align(po.aligner_params(10), [sp], [scan], [wl.map_points], wl.x0[i].astype(np.float64), double=True)
            assert r["status"] == want["status"] and np.allclose(r["pose"], want["pose_after_10_its_fp64"], atol=1e-6)
            assert abs(r["stats"][0].n_corr - want["n_corr_first"]) <= 2 and abs(r["stats"][0].chi_in - want["chi_first"]) <= 1e-3 * max(want["chi_first"], 1.0)


def test_fixed_sincos_accuracy_and_quadrants(po):
    """lsmo_sincosf replaces libm's cosf / sinf for pose rotations (same operation sequence on the GPU): within 1.2e-7 of
    float64 sin / cos over the angles poses take, exact at 0, right signs in every quadrant, odd / even symmetry."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-7, 7, 20000), np.linspace(-math.pi, math.pi, 4001), [0.0, math.pi / 2, -math.pi / 2, math.pi, 100.0, -2500.0]]).astype(np.float32)
    s, c = po.sincos(x)
    assert np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64))).max() < 1.2e-7
    assert np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64))).max() < 1.2e-7
    s0, c0 = po.sincos([0.0]); assert s0[0] == 0.0 and c0[0] == 1.0
    sp, cp = po.sincos(x); sm, cm = po.sincos(-x)
    assert np.array_equal(sm, -sp) and np.array_equal(cm, cp)
    assert np.abs(s * s + c * c - 1.0).max() < 3e-7


def test_device_order_summation_is_the_same_sum(po):
    """lsmo_aligner_params.device_order re-associates the per-pair sums (thread = column mod 512, DPP scan tree, waves in order) and
    changes nothing else: first-iteration statistics are identical up to fp32 summation noise, poses agree to ~1e-5, for every
    finder, with Cauchy and with a prior."""
    wl = synth.make_workload(3, 20000, seed=4)
    scans = [wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]] for i in range(3)]
    cases = [po.slice_params(), po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=0.02),
             po.slice_params(finder=po.FINDER_NN, max_distance=0.3), po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.1)]
    for sp in cases:
        for i in range(3):
            kw = dict(prior_z=wl.x0[i], prior_omega=np.eye(3) * 10.0) if i == 2 else {}
            a = po.align(po.aligner_params(10, **kw), [sp], [scans[i]], [wl.map_points], wl.x0[i])
            b = po.align(po.aligner_params(10, device_order=True, **kw), [sp], [scans[i]], [wl.map_points], wl.x0[i])
            assert a["status"] == b["status"] == 0 and a["iterations"] == b["iterations"]
            sa, sb = a["stats"][0], b["stats"][0]
            assert (sa.n_corr, sa.n_in, sa.n_out) == (sb.n_corr, sb.n_in, sb.n_out)
            assert abs(sa.chi_in - sb.chi_in) <= 2e-5 * abs(sa.chi_in) + 1e-9
            d = np.abs(a["pose"] - b["pose"])
            assert d.max() < 2e-5, (sp.finder, i, d)
            assert np.abs(a["H"] - b["H"]).max() <= 2e-4 * np.abs(a["H"]).max()


def test_fixed_log_accuracy(po):
    """lsmo_logf_fixed replaces libm's logf in the Cauchy kernel statistic (same operation sequence on the GPU)."""
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(1.0, 8.0, 20000), np.exp(rng.uniform(0.0, 60.0, 20000)), [1.0, 2.0, 1.4142135, 1.4142137]]).astype(np.float32)
    got = po.log_fixed(x).astype(np.float64); ref = np.log(x.astype(np.float64))
    assert (np.abs(got - ref) / np.maximum(ref, 0.1)).max() < 2e-7
    assert po.log_fixed([1.0])[0] == 0.0


def test_tracker_chain_digests(po):
    """tests/golden/tracker_chain.json: an 8-step live-tracker chain (ranges -> preprocess -> clip -> two-slice align with prior, the
    kernels' summation order -> merge) reduced to digests of every intermediate array.  The oracle must reproduce the committed file
    bit for bit; tests/test_gpu_configs_at_size.py holds the HIP path to the same file."""
    import tracker_chain
    g = json.load(open(golden_path("tracker_chain.json")))
    got = tracker_chain.run_oracle(po, len(g["steps"]))
    assert got == g["steps"]
    assert all(st["status"] == 0 for st in got) and got[-1]["map_points"] > got[0]["map_points"]



def test_termination_chi_epsilon_semantics(po, small_workload):
    """lsmo_aligner_params.termination_chi_epsilon (the aligner's "termination_criteria", MULTI.json:627-630: unset in the shipped aligners):
    0 runs max_iterations; epsilon > 0 stops after the first iteration whose total chi^2 differs from the previous one's by less than
    epsilon times itself -- that iteration is still solved and applied; a looser epsilon never stops later; the pose at the stop is the pose
    of the full run's same iteration."""
    wl = small_workload
    s = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    full = po.align(po.aligner_params(20), [po.slice_params()], [s], [wl.map_points], wl.x0[0])
    assert full["iterations"] == 20
    its = []
    for eps in (1e-4, 1e-2, 0.5):
        r = po.align(po.aligner_params(20, termination_chi_epsilon=eps), [po.slice_params()], [s], [wl.map_points], wl.x0[0])
        assert r["status"] == 0 and 2 <= r["iterations"] <= 20
        k = r["iterations"]
        chi = [st.chi_in + st.chi_out for st in full["stats"]]
        assert abs(np.float32(chi[k - 2]) - np.float32(chi[k - 1])) < eps * chi[k - 1]             # the stopping iteration satisfies the criterion ...
        assert all(not (abs(np.float32(chi[j - 1]) - np.float32(chi[j])) < np.float32(eps) * np.float32(chi[j])) for j in range(1, k - 1))      # ... and no earlier one did
        short = po.align(po.aligner_params(k), [po.slice_params()], [s], [wl.map_points], wl.x0[0])
        assert np.array_equal(short["pose"], r["pose"]) and np.array_equal(short["H"], r["H"])
        its.append(k)
    assert its[0] >= its[1] >= its[2] >= 2 and its[0] < 20


def test_kdtree_oracle_is_approximate_and_honours_its_parameters(po, small_workload):
    """The believed upstream tree: never a nearer neighbour than the exact search, sometimes a farther one or none; a leaf
    range above the cloud's extent (one leaf holding everything) makes it the exact search; min_leaf_points changes the pairs."""
    wl = small_workload
    s = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    x = wl.x0[0]
    ex = po.find(po.slice_params(finder=po.FINDER_NN, max_distance=0.5, normal_cos=-1.0), s, wl.map_points[::7], x)
    kd = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5, normal_cos=-1.0), s, wl.map_points[::7], x)
    one_leaf = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5, normal_cos=-1.0, kd_max_leaf_range=1e3), s, wl.map_points[::7], x)
    other = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5, normal_cos=-1.0, kd_min_leaf_points=4), s, wl.map_points[::7], x)
    assert len(kd) <= len(ex) and not np.array_equal(kd, ex) and not np.array_equal(kd, other)
    # one leaf: every query scans every point; strict '<' against max_distance^2 (exact search: '<=') -- same pairs unless a pair sits exactly on the gate
    assert np.array_equal(one_leaf, ex)


def test_tracker_replay_1000_digests(po):
    """tests/golden/tracker_replay_1000.json (BASELINE configs[2] at its size: 1 000 steps of the MULTI-parameter tracker chain, digests of
    every 50th step): the oracle must reproduce the committed file bit for bit; tests/test_gpu_configs_at_size.py holds the HIP path to it."""
    import tracker_chain
    g = json.load(open(golden_path("tracker_replay_1000.json")))
    got = tracker_chain.run_oracle(po, g["steps_total"], record_every=g["record_every"])
    assert got == g["steps"]
    assert len(got) == 20 and all(st["status"] == 0 for st in got)


def test_pair_digest_definitions_agree_and_tell_sets_apart(po, small_workload):
    """lsm2d_iteration_stats.pair_digest: the C oracle's hash, its numpy restatement and the product library's lsm2d_pair_hash (host code, no GPU
    needed) are the same function; the digest of a correspondence set does not depend on the order of its pairs, and exchanging partners,
    moving one index by one or swapping the slice changes it."""
    from srrg2_laser_slam_2d_amd import _capi
    L = po.lib(); lib = _capi.load()
    rng = np.random.default_rng(5)
    p = rng.integers(0, 2 ** 31 - 1, size=(2000, 2)); p[:50] = rng.integers(0, 40, size=(50, 2))
    for sl in (0, 1, 2, 3):
        c = 0
        for f, m in p[:300]:
            h = L.lsmo_pair_hash(sl, int(f), int(m))
            assert h == lib.lsm2d_pair_hash(sl, int(f), int(m))
            c = (c + h) & 0xFFFFFFFFFFFFFFFF
        assert c == po.pair_digest(p[:300], sl)
    base = po.pair_digest(p)
    assert base == po.pair_digest(p[rng.permutation(len(p))])                      # order-independent
    q = p.copy(); q[[10, 11], 1] = q[[11, 10], 1]; assert po.pair_digest(q) != base      # partners exchanged
    q = p.copy(); q[7, 0] += 1; assert po.pair_digest(q) != base
    q = p.copy(); q[7, 1] += 1; assert po.pair_digest(q) != base
    assert po.pair_digest(p, 1) != base and po.pair_digest(p[:-1]) != base
    # the aligner's statistics carry it: every iteration's digest is the digest of the finder's pairs at that iteration's pose, both summation orders
    wl = small_workload
    s = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    for dev in (False, True):
        r = po.align(po.aligner_params(6, device_order=dev), [po.slice_params()], [s], [wl.map_points], wl.x0[1], want_pairs=True)
        assert r["stats"][0].pair_digest == po.pair_digest(po.find(po.slice_params(), s, wl.map_points, wl.x0[1]))
        assert r["stats"][-1].pair_digest == po.pair_digest(r["pairs"][0]) and len(r["pairs"][0]) == r["stats"][-1].n_corr
    # two slices: the second slice's pairs are salted with its index
    r2 = po.align(po.aligner_params(3), [po.slice_params(), po.slice_params(normal_cos=0.9)], [s, s], [wl.map_points, wl.map_points], wl.x0[1], want_pairs=True)
    assert r2["stats"][-1].pair_digest == (po.pair_digest(r2["pairs"][0], 0) + po.pair_digest(r2["pairs"][1], 1)) & 0xFFFFFFFFFFFFFFFF
    assert r2["stats"][-1].n_corr == len(r2["pairs"][0]) + len(r2["pairs"][1])


def test_inlier_only_runs_and_keep_only_inlier_correspondences_semantics(po, small_workload):
    """MultiAligner2D's two remaining options as restated in lsm2d_oracle.h (MULTI.json:606-610; [UPSTREAM-MEMORY]):
    enable_inlier_only_runs -- a second loop of up to max_iterations iterations when the regular loop ended well with enough inliers, in which
    non-inliers weigh nothing and inliers 1; keep_only_inlier_correspondences -- the pairs handed back are the last iteration's inliers."""
    wl = small_workload
    s = wl.scan_points[wl.scan_offsets[2]:wl.scan_offsets[3]].copy()
    s[:, :2] += np.random.default_rng(11).normal(0.0, 0.02, (len(s), 2)).astype(np.float32)      # range noise: outliers under a tight kernel, to the end
    x_off = wl.x0[2] + np.float32([0.15, -0.1, 0.05])
    tau = 5e-4
    spc = po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=tau)
    for mode in (False, True, "ref"):
        dt = np.float64 if mode is True else np.float32
        kw = dict(double=mode)
        plain = po.align(po.aligner_params(8), [spc], [s], [wl.map_points], x_off.astype(dt), want_pairs=True, **kw)
        runs = po.align(po.aligner_params(8, enable_inlier_only_runs=True), [spc], [s], [wl.map_points], x_off.astype(dt), want_pairs=True, **kw)
        assert plain["iterations"] == 8 and runs["iterations"] == 16 and runs["status"] == 0
        # the regular loop is untouched by the option: its eight iterations are the plain run's, bit for bit
        for k in range(8):
            a, b = plain["stats"][k], runs["stats"][k]
            assert (a.n_corr, a.n_in, a.n_out, a.chi_in, a.chi_out, a.pair_digest) == (b.n_corr, b.n_in, b.n_out, b.chi_in, b.chi_out, b.pair_digest)
        # not enough inliers at the end of the regular loop: no second loop
        few = po.align(po.aligner_params(8, min_num_inliers=10 ** 6, enable_inlier_only_runs=True), [spc], [s], [wl.map_points], x_off.astype(dt), **kw)
        assert few["iterations"] == 8 and few["status"] == po.NOT_ENOUGH_INLIERS
        # keep_only_inlier_correspondences: a subset of the unfiltered vector, as many as the last iteration's inliers; nothing else moves
        keep = po.align(po.aligner_params(8, keep_only_inlier_correspondences=True), [spc], [s], [wl.map_points], x_off.astype(dt), want_pairs=True, **kw)
        assert np.array_equal(keep["pose"], plain["pose"]) and np.array_equal(keep["H"], plain["H"])
        assert len(keep["pairs"][0]) == plain["stats"][-1].n_in and plain["stats"][-1].n_out > 0
        assert {tuple(p) for p in keep["pairs"][0].tolist()} < {tuple(p) for p in plain["pairs"][0].tolist()}
    # a slice without robustifier has no outliers: the second loop is eight more regular iterations
    sp = po.slice_params()
    r16 = po.align(po.aligner_params(16), [sp], [s], [wl.map_points], wl.x0[2])
    r8x2 = po.align(po.aligner_params(8, enable_inlier_only_runs=True), [sp], [s], [wl.map_points], wl.x0[2])
    assert r8x2["iterations"] == 16 and np.array_equal(r16["pose"], r8x2["pose"]) and np.array_equal(r16["H"], r8x2["H"])
    # the weights of the second loop by hand: H of its first iteration = sum over the INLIERS of J^T J at that iteration's pose (fp64 oracle)
    runs = po.align(po.aligner_params(8, enable_inlier_only_runs=True), [spc], [s], [wl.map_points], x_off.astype(np.float64), double=True)
    nine = po.align(po.aligner_params(1, enable_inlier_only_runs=True, min_num_inliers=0), [spc], [s], [wl.map_points], x_off.astype(np.float64), double=True)
    assert nine["iterations"] == 2
    first = po.align(po.aligner_params(1), [spc], [s], [wl.map_points], x_off.astype(np.float64), double=True)
    pairs = po.find(spc, s, wl.map_points, first["pose"], double=True)
    H = np.zeros((3, 3))
    for fi, mi in pairs:
        e, J = po.error_jacobian(s[fi], wl.map_points[mi], first["pose"])
        if float(e @ e) < tau:
            H += J.T @ J
    assert np.allclose(nine["H"], H, rtol=1e-9, atol=1e-9)
    # the termination criterion starts afresh in the second loop
    te = po.align(po.aligner_params(12, termination_chi_epsilon=1e-2, enable_inlier_only_runs=True), [spc], [s], [wl.map_points], x_off)
    k1 = po.align(po.aligner_params(12, termination_chi_epsilon=1e-2), [spc], [s], [wl.map_points], x_off)["iterations"]
    assert k1 < 12 and k1 + 2 <= te["iterations"] <= k1 + 12


def test_fuzz_case_generators_reproduce_a_logged_soak_case():
    """tests/fuzz_cases.py regenerates any trial of the GPU fuzz tests from (test, seed, trial) alone; the round-5 soak logged the device's pose for its
    violators (profiles/r05/fuzz_soak_r05r_*.log).  The device equals the device-order mirror bit for bit, so the mirror run on the regenerated inputs must
    give the logged bits -- which pins the generators' draw order (what tests/replay_violators.py and the named exception lists rely on)."""
    import replay_violators as rv
    test, seed, trial, ali, logged, _ = rv.CASES[3]      # structure fuzz, seed 17, trial 346, alignment 0 (a 4000-point map: quick)
    osl, fx, mv, kw, x0, _ = rv.inputs_of(test, seed, trial, ali)
    from oracle import pyoracle as po
    rt = po.align(po.aligner_params(device_order=True, **kw), osl, fx, mv, x0)
    assert np.array_equal(np.asarray(logged, np.float32), rt["pose"])
    r = po.align(po.aligner_params(**kw), osl, fx, mv, x0); rd = po.align(po.aligner_params(**kw), osl, fx, mv, x0.astype(np.float64), double=True)
    assert np.abs(r["pose"] - rd["pose"])[:2].max() < 1e-5      # the sequential order (what "sum_order" 1 computes) sits on the fp64 oracle here
    assert (test, seed, trial, ali) in fuzz_cases_known()


def fuzz_cases_known():
    import fuzz_cases
    return fuzz_cases.KNOWN_TREE_ORDER_DEVIATIONS
