"""GPU parity tests, rows a6-a10: the factor, the robustifier, the Gauss-Newton step and the aligner loop with its options, against the oracle in all its arithmetic modes.

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


def test_factor_known_answer_and_parity(ctx, po, small_workload):
    g = json.load(open(golden_path("nicp_2d_known_answer.json")))
    fixed = np.array(g["fixed"], np.float32); moving = np.array(g["moving"], np.float32)
    corr = np.array([[0, 0], [1, 1], [2, 2]], np.int32)
    sp = api.make_slice_params()
    H, b, st = api.linearize(ctx, sp, fixed, moving, corr, g["pose"])
    assert np.allclose(H, g["H"], atol=2e-5) and np.allclose(b, g["b"], atol=2e-5)
    assert st.n_correspondences == 3 and st.n_inliers == 3 and abs(st.chi_inliers - g["chi"]) < 1e-5
    spc = api.make_slice_params(robustifier=api.ROBUST_CAUCHY, chi_threshold=g["cauchy"]["tau"])
    Hc, bc, stc = api.linearize(ctx, spc, fixed, moving, corr, g["pose"])
    assert np.allclose(Hc, g["cauchy"]["H"], atol=2e-5) and np.allclose(bc, g["cauchy"]["b"], atol=2e-5)
    assert stc.n_inliers == g["cauchy"]["n_inliers"] and abs(stc.chi_outliers - g["cauchy"]["chi_outliers"]) < 1e-5
    # a real correspondence set from the finder, against the fp64 oracle
    wl = small_workload
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    osp = po.slice_params()
    corr = po.find(osp, f, wl.map_points, wl.x0[0])
    for robust in (api.ROBUST_NONE, api.ROBUST_CAUCHY):
        sp = api.make_slice_params(robustifier=robust, chi_threshold=0.05)
        H, b, st = api.linearize(ctx, sp, f, wl.map_points, corr, wl.x0[0])
        oH, ob, ost = po.linearize(po.slice_params(robustifier=robust, chi_threshold=0.05), f, wl.map_points, corr, wl.x0[0].astype(np.float64), double=True)
        assert np.allclose(H, oH, rtol=2e-5, atol=2e-5 * np.abs(oH).max())
        assert np.allclose(b, ob, rtol=2e-5, atol=2e-5 * max(np.abs(ob).max(), 1.0))
        assert st.n_correspondences == len(corr) and st.n_inliers == ost.n_in and st.n_outliers == ost.n_out
        assert abs(st.chi_inliers - ost.chi_in) <= 2e-5 * max(ost.chi_in, 1.0)
        assert abs(st.chi_outliers - ost.chi_out) <= 2e-5 * max(ost.chi_out, 1.0)
        # ... and BITWISE against the fp32 oracle summing in this launch's order (a few hundred pairs: two workgroups of 256)
        tH, tb, tst = po.linearize_device_order(po.slice_params(robustifier=robust, chi_threshold=0.05), f, wl.map_points, corr, wl.x0[0])
        assert np.array_equal(H, tH) and np.array_equal(b, tb) and np.float32(st.chi_inliers) == np.float32(tst.chi_in)
    # the NN finder's thousands of pairs: many workgroups, the launch's two-level order
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=0.3)
    corr = po.find(osp, f, wl.map_points, wl.x0[0])
    assert len(corr) > 3000
    H, b, st = api.linearize(ctx, api.make_slice_params(), f, wl.map_points, corr, wl.x0[0])
    tH, tb, tst = po.linearize_device_order(po.slice_params(), f, wl.map_points, corr, wl.x0[0])
    assert np.array_equal(H, tH) and np.array_equal(b, tb) and np.float32(st.chi_inliers) == np.float32(tst.chi_in) and st.n_correspondences == len(corr)
    # empty correspondence vector
    H, b, st = api.linearize(ctx, sp, f, wl.map_points, np.zeros((0, 2), np.int32), wl.x0[0])
    assert np.all(H == 0) and np.all(b == 0) and st.n_correspondences == 0


def test_aligner_single_reference_usage(ctx, po, small_workload):
    """apps/visual_test_aligner_2d.cpp:123-156 with fixed = scan, moving = local map."""
    wl = small_workload
    al = _aligner(ctx)
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    al.setFixed({"points": f}); al.setMoving({"points": wl.map_points}); al.setMovingInFixed(wl.x0[0])
    assert al.compute() == 0
    r = po.align(po.aligner_params(20), [po.slice_params()], [f], [wl.map_points], wl.x0[0])
    rd = po.align(po.aligner_params(20), [po.slice_params()], [f], [wl.map_points], wl.x0[0].astype(np.float64), double=True)
    for ref in (r["pose"], rd["pose"], wl.x_true[0]):
        d = np.abs(al.movingInFixed() - ref)
        assert d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    st = al.iterationStats()
    assert len(st) == 20 and st["n_correspondences"][0] == r["stats"][0].n_corr      # first iteration: same pose, same pairs
    assert abs(st["chi_inliers"][0] - r["stats"][0].chi_in) <= 1e-4 * r["stats"][0].chi_in
    assert np.allclose(al.informationMatrix(), rd["H"], rtol=1e-3, atol=1e-3 * np.abs(rd["H"]).max())


def test_aligner_batch_matches_oracle_and_truth(ctx, po):
    wl = synth.make_workload(48, 100000, seed=1)
    al = _aligner(ctx)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    res = al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
    xo, _, status, _ = po.align_batch(po.aligner_params(20), po.slice_params(), wl.scan_points, wl.scan_offsets, wl.map_points, wl.x0, n_threads=8)
    assert np.all(res.status == 0) and np.all(status == 0) and np.all(res.iterations == 20)
    d = np.abs(res.pose - xo)
    assert d[:, :2].max() < POSE_TOL_M and d[:, 2].max() < POSE_TOL_RAD
    dt = np.abs(res.pose - wl.x_true)
    assert dt[:, :2].max() < POSE_TOL_M and dt[:, 2].max() < POSE_TOL_RAD
    # bitwise reproducible run to run (z-buffer min and fixed-order reductions are order independent)
    res2 = al.compute_batch([fixed], [moving], wl.x0)
    assert np.array_equal(res.pose, res2.pose) and np.array_equal(res.information, res2.information)
    # and bit-identical to the fp32 oracle summing in the kernels' order: full size (100k-point map, 20 iterations), every 6th alignment
    for i in range(0, 48, 6):
        rt = po.align(po.aligner_params(20, device_order=True), [po.slice_params()], [wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]], [wl.map_points], wl.x0[i])
        _assert_bitwise_equal_to_device_order_oracle(res, i, rt, ("batch", i))


def test_aligner_noisy_data_and_cauchy(ctx, po):
    wl = synth.make_workload(16, 50000, seed=4, map_noise=0.01, scan_noise=0.01)
    for rb in (None, api.RobustifierCauchy(0.05)):
        al = _aligner(ctx, robustifier=rb)
        fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
        res = al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
        osp = po.slice_params(robustifier=po.ROBUST_CAUCHY if rb else po.ROBUST_NONE, chi_threshold=0.05)
        xo, _, status, last = po.align_batch(po.aligner_params(20), osp, wl.scan_points, wl.scan_offsets, wl.map_points, wl.x0, n_threads=8)
        assert np.array_equal(res.status, status)
        d = np.abs(res.pose - xo)
        assert d[:, :2].max() < POSE_TOL_M and d[:, 2].max() < POSE_TOL_RAD
        for i in range(0, 16, 3):           # noisy data, Cauchy: still the mirror's bits in the kernels' summation order
            osp_t = po.slice_params(robustifier=po.ROBUST_CAUCHY if rb else po.ROBUST_NONE, chi_threshold=0.05)
            rt = po.align(po.aligner_params(20, device_order=True), [osp_t], [wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]], [wl.map_points], wl.x0[i])
            _assert_bitwise_equal_to_device_order_oracle(res, i, rt, ("noisy", bool(rb), i))


def test_aligner_status_codes_and_ragged_inputs(ctx, po, small_workload):
    wl = small_workload
    n = len(wl.x0)
    # alignment 1 gets a hopeless initial guess, alignment 2 an empty scan
    offs = wl.scan_offsets.copy()
    pts = np.concatenate([wl.scan_points[:offs[2]], wl.scan_points[offs[3]:]], 0)
    offs[3:] -= (offs[3] - offs[2])
    x0 = wl.x0.copy(); x0[1] += np.float32([80, 80, 0])
    al = _aligner(ctx)
    fixed = api.CloudSet(ctx, pts, offs); moving = api.CloudSet(ctx, wl.map_points)
    res = al.compute_batch([fixed], [moving], x0, want_stats=True)
    xo, _, status, _ = po.align_batch(po.aligner_params(20), po.slice_params(), pts, offs, wl.map_points, x0)
    assert np.array_equal(res.status, status)
    assert res.status[1] == 1 and res.status[2] == 1 and res.iterations[1] == 1
    assert np.array_equal(res.pose[1], x0[1]) and np.array_equal(res.pose[2], x0[2])
    ok = res.status == 0
    assert ok.sum() == n - 2 and np.abs(res.pose[ok] - xo[ok]).max() < POSE_TOL_M
    # NotEnoughInliers
    al2 = _aligner(ctx); al2.param_min_num_inliers = 100000
    assert np.all(al2.compute_batch([fixed], [moving], wl.x0).status[[0, 3]] == 2)
    # SingularH: one wall only
    wall = np.stack([np.linspace(-3, 3, 400), np.full(400, 2.0), np.zeros(400), -np.ones(400)], 1).astype(np.float32)
    al3 = _aligner(ctx, 360); al3.param_slice_processors[0].param_min_num_correspondences = 0
    al3.setFixed({"points": wall}); al3.setMoving({"points": wall}); al3.setMovingInFixed([0, 0, 0])
    assert al3.compute() == 3
    # zero iterations
    al4 = _aligner(ctx, its=0)
    r4 = al4.compute_batch([fixed], [moving], wl.x0)
    assert np.all(r4.status == 0) and np.array_equal(r4.pose, wl.x0) and np.all(r4.iterations == 0)


def test_aligner_cloud_index_selection(ctx, small_workload):
    """loop-closure style: candidates pick their scan through an index array; one shared map."""
    wl = small_workload
    al = _aligner(ctx, its=10)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    base = al.compute_batch([fixed], [moving], wl.x0)
    perm = np.array([3, 0, 5, 5, 1], np.int32)
    res = al.compute_batch([fixed], [moving], wl.x0[perm], fixed_index=perm[None, :])
    assert np.array_equal(res.pose, base.pose[perm])


def test_ragged_moving_clouds_through_the_lane_chunked_stream(ctx, po, small_workload):
    """k_align streams a moving cloud from its lane-chunked copy in steps of one pair per thread, two steps per trip.  A set
    mixing every step count that matters (0, 1, 2, 3 and more, odd and even sizes, exactly / just over a multiple of 512
    pairs) goes through one launch, each cloud chosen by an index array; status, iteration count and pose equal the oracle's
    for every cloud, and the correspondence counts of the first iteration are equal (bit-exact z-buffers)."""
    wl = small_workload
    rng = np.random.default_rng(5)
    sizes = [0, 1, 2, 7, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 3073, 4096, 4099, 5121, 12001]
    perm = rng.permutation(len(wl.map_points))
    clouds = [wl.map_points[np.sort(perm[:k])] for k in sizes]          # subsets of the map, in map order
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    moving = api.CloudSet(ctx, np.concatenate(clouds, 0), offs)
    scan = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    al = _aligner(ctx, its=8)
    n = len(sizes)
    x0 = np.repeat(wl.x0[:1], n, 0)
    res = al.compute_batch([api.CloudSet(ctx, scan)], [moving], x0, moving_index=np.arange(n, dtype=np.int32)[None, :], want_stats=True)
    for i, c in enumerate(clouds):
        r = po.align(po.aligner_params(8), [po.slice_params()], [scan], [c], wl.x0[0])
        assert res.status[i] == r["status"] and res.iterations[i] == r["iterations"], (sizes[i], res.status[i], r["status"])
        assert res.stats[i]["n_correspondences"][0] == r["stats"][0].n_corr, sizes[i]
        d = np.abs(res.pose[i] - r["pose"])
        assert d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD, (sizes[i], d)
    assert (res.status == 0).sum() >= 8 and (res.status == 1).sum() >= 3


def test_multi_slice_sensor_offsets_and_prior(ctx, po):
    world = synth.make_world(5)
    m = synth.make_map(world, 30000)
    robot = synth.sample_poses(world, 1, seed=11)
    S0, S1 = np.array([0.2, 0.1, 0.1]), np.array([-0.3, 0.0, math.pi])
    scans = [synth.make_scans(world, synth.compose_poses(robot, S[None, :]), n_beams=721)[0] for S in (S0, S1)]
    x0 = synth.invert_poses(synth.compose_poses(robot, np.array([[0.04, -0.03, 0.03]])))[0].astype(np.float32)
    # two different projectors (columns AND range gate): the slices share one moving canvas inside the kernel
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    proj1 = api.PointNormal2fProjectorPolar(541, -math.pi, math.pi, 0.5, 9.0)
    al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.9), sensor_in_robot=S0, robustifier=api.RobustifierCauchy(0.01),
        min_num_correspondences=5, fixed_slice_name="points_0", moving_slice_name="points"))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
        api.CorrespondenceFinderProjective2f(ctx, proj1, 0.5, 0.8), sensor_in_robot=S1, min_num_correspondences=5,
        fixed_slice_name="points_1", moving_slice_name="points"))
    al.setFixed({"points_0": scans[0], "points_1": scans[1]}); al.setMoving({"points": m}); al.setMovingInFixed(x0)
    osl = [_oracle_slice(po, s.slice_params()) for s in al.param_slice_processors]
    for prior in (None, (x0, np.eye(3, dtype=np.float32) * 50.0)):
        al._prior = None
        if prior:
            al.setPrior(*prior)
        assert al.compute() == 0
        ap = po.aligner_params(10, prior_z=prior[0] if prior else None, prior_omega=prior[1] if prior else None)
        r = po.align(ap, osl, scans, [m, m], x0.astype(np.float64), double=True)
        d = np.abs(al.movingInFixed() - r["pose"])
        assert r["status"] == 0 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
        assert al.iterationStats()["n_correspondences"][-1] == r["stats"][-1].n_corr


def test_aligner_nn_role_b_scan_queries_map(ctx, po):
    """BASELINE wording: search structure over the local map, scans as queries (fixed = map, moving = scan)."""
    wl = synth.make_workload(24, 100000, seed=6)
    x0_b = synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32); xt_b = synth.invert_poses(wl.x_true)
    al = _nn_aligner(ctx)
    fixed = api.CloudSet(ctx, wl.map_points); moving = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    res = al.compute_batch([fixed], [moving], x0_b, want_stats=True)
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=0.5)
    for i in range(0, 24, 4):
        s = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        r = po.align(po.aligner_params(20), [osp], [wl.map_points], [s], x0_b[i])
        d = np.abs(res.pose[i] - r["pose"])
        assert res.status[i] == r["status"] == 0 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
        assert res.stats[i]["n_correspondences"][0] == r["stats"][0].n_corr
    # the map points are ~2 mm apart, so point-to-plane NN ICP lands within a few mm of the generating pose
    dt = np.abs(res.pose - xt_b)
    assert dt[:, :2].max() < 5e-3 and dt[:, 2].max() < 2e-3


def test_aligner_nn_role_a_map_queries_scan(ctx, po):
    """reference tracker wiring: tree over the scan, every map point is a query (up to N_m correspondences)."""
    wl = synth.make_workload(8, 30000, seed=7)
    al = _nn_aligner(ctx, md=0.3)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    res = al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=0.3)
    for i in range(0, 8, 3):
        s = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        r = po.align(po.aligner_params(20), [osp], [s], [wl.map_points], wl.x0[i])
        d = np.abs(res.pose[i] - r["pose"])
        assert res.status[i] == r["status"] and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
        assert res.stats[i]["n_correspondences"][0] == r["stats"][0].n_corr


def test_mixed_finders_two_slices(ctx, po, small_workload):
    """one projective slice + one NN slice sharing the pose (exercises the k_align<true,true> instantiation)."""
    wl = small_workload
    s = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector()), min_num_correspondences=10))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3), min_num_correspondences=10))
    al.setFixed({"points": s}); al.setMoving({"points": wl.map_points}); al.setMovingInFixed(wl.x0[0])
    assert al.compute() == 0
    osl = [po.slice_params(), po.slice_params(finder=po.FINDER_NN, max_distance=0.3)]
    r = po.align(po.aligner_params(10), osl, [s, s], [wl.map_points, wl.map_points], wl.x0[0])
    d = np.abs(al.movingInFixed() - r["pose"])
    assert r["status"] == 0 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    assert al.iterationStats()["n_correspondences"][0] == r["stats"][0].n_corr


def test_hip_path_against_the_reference_arithmetic_mode(ctx, po):
    """The HIP path (fixed-polynomial atan2 / sin / cos / log, fused multiply-adds, tree sums) against the oracle in the REFERENCE'S
    OWN ARITHMETIC (`_r`: libm, no FMA, Eigen's association, sums pair after pair -- oracle/lsm2d_oracle.h): poses within the
    north_star tolerance on BASELINE configs[1], [3] and [4], with the fraction of first-iteration pairs that differ reported
    (PARITY.md section 5 holds the full table)."""
    cases = (("configs[1]", 100000, 16, 0.0, 0), ("configs[3] Cauchy", 100000, 8, 0.05, 3), ("configs[4]", 1000000, 3, 0.0, 5))
    proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
    for name, n_map, n, tau, seed in cases:
        wl = synth.make_workload(n, n_map, seed=seed)
        al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
        finder = api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, robustifier=api.RobustifierCauchy(tau) if tau > 0 else None,
                                                                          min_num_correspondences=10))
        fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
        res = al.compute_batch([fixed], [moving], wl.x0)
        osp = po.slice_params(**({"robustifier": po.ROBUST_CAUCHY, "chi_threshold": tau} if tau > 0 else {}))
        worst = np.zeros(2); differing = []
        for i in range(n):
            sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
            ref = po.align(po.aligner_params(20), [osp], [sc], [wl.map_points], wl.x0[i], double="ref")
            assert ref["status"] == 0 and res.status[i] == 0
            d = np.abs(res.pose[i] - ref["pose"]); worst = np.maximum(worst, [d[:2].max(), d[2]])
            finder.setFixed(fixed, i); finder.setMoving(moving); finder.setLocalMapInSensor(wl.x0[i])
            got = {tuple(p) for p in finder.compute().tolist()}
            want = {tuple(p) for p in po.find(osp, sc, wl.map_points, wl.x0[i], double="ref").tolist()}
            differing.append(len(got ^ want) / max(len(got | want), 1))
        print("%s: HIP vs reference arithmetic: max pose delta %.2e m / %.2e rad, first-iteration pairs differing %.3f %% (mean over %d scans)"
              % (name, worst[0], worst[1], 100 * np.mean(differing), n))
        assert worst[0] < 1e-4 and worst[1] < 1e-4, (name, worst)
        assert np.mean(differing) < 0.08, (name, differing)
        fixed.close(); moving.close()


def test_maximum_sizes_against_oracle(ctx, po):
    """BASELINE configs[4] scale in a unit test: a 1M-point local map (oracle: ~0.2 s per alignment), plus the widest
    scan the preprocessor takes (2048 beams) and a ragged batch with single-point and odd-sized clouds."""
    world = synth.make_world(0)
    m = synth.make_map(world, 1_000_000)
    poses = synth.sample_poses(world, 3, seed=17)
    scans, offs = synth.make_scans(world, poses)
    x_true, x0 = synth.initial_guesses(poses, seed=17)
    al = _aligner(ctx)
    res = al.compute_batch([api.CloudSet(ctx, scans, offs)], [api.CloudSet(ctx, m)], x0.astype(np.float32), want_stats=True)
    xo, _, status, last = po.align_batch(po.aligner_params(20), po.slice_params(), scans, offs, m, x0.astype(np.float32), n_threads=3)
    d = np.abs(res.pose - xo)
    assert np.array_equal(res.status, status) and d[:, :2].max() < POSE_TOL_M and d[:, 2].max() < POSE_TOL_RAD
    assert [int(s) for s in res.last_stats()["n_correspondences"]] == [l.n_corr for l in last]
    assert np.abs(res.pose - x_true)[:, :2].max() < POSE_TOL_M
    # finder level on the 1M map: bit-exact pairs
    f = api.CorrespondenceFinderProjective2f(ctx, _projector())
    f.setFixed(scans[offs[0]:offs[1]]); f.setMoving(m); f.setLocalMapInSensor(x0[0].astype(np.float32))
    assert np.array_equal(f.compute(), po.find(po.slice_params(), scans[offs[0]:offs[1]], m, x0[0].astype(np.float32)))
    # widest scan of the preprocessor
    rng = synth.make_scan_ranges(world, poses, n_beams=2048, angle_min=-math.pi, angle_max=math.pi)
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=30.0, voxelize_resolution=0.02)
    pre.setRawData(rng, -math.pi, math.pi)
    meas = pre.compute()
    pp = po.Preprocessor(2048, -math.pi, math.pi, 0.3, 30.0, 0.3, 5, 0.02)
    for i in range(3):
        assert np.array_equal(meas.download(i), po.preprocess_scan(pp, rng[i]))
    with pytest.raises(api.Lsm2dError):
        pre.setRawData(np.ones((1, 2049), np.float32), -1.0, 1.0); pre.compute()
    # ragged set: clouds of 1, 2, 3, 1081 and 0 points (odd sizes exercise the even-aligned starts)
    c = scans[offs[0]:offs[1]]
    ragged = np.concatenate([c[:1], c[:2], c[:3], c, c[:0]], 0); roffs = np.array([0, 1, 3, 6, 6 + len(c), 6 + len(c)], np.int32)
    rs = api.CloudSet(ctx, ragged, roffs)
    for i, want in enumerate((c[:1], c[:2], c[:3], c, c[:0])):
        assert np.array_equal(rs.download(i), want)
    r5 = al.compute_batch([rs], [api.CloudSet(ctx, m)], np.tile(x0[0].astype(np.float32), (5, 1)))
    assert list(r5.status) == [1, 1, 1, 0, 1] and np.array_equal(r5.pose[3], res.pose[0])


def test_aligner_kdtree_both_roles_bitwise(ctx, po):
    """k_align with the KD-tree finder fused in: status, iterations, pose, information matrix and every iteration's statistics BITWISE equal to
    the oracle running the believed upstream tree and summing in the device's order -- role B (tree over the 100k-point map, scans as queries:
    BASELINE's wording) and role A (a tree per scan, every map point a query: the reference tracker's wiring), with and without the top of
    the tree staged in LDS, with the Cauchy kernel."""
    wl = synth.make_workload(12, 100000, seed=6)
    x0_b = synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32); xt_b = synth.invert_poses(wl.x_true)
    fixed = api.CloudSet(ctx, wl.map_points); moving = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    results = []
    for lds_nodes, modes in ((1024, 1), (0, 1), (37, 1), (1024, 0)):      # (modes 0: the shared instantiation with both forms of the descent -- experiments build)
        if xset(ctx, kd_lds_nodes=lds_nodes, kd_modes=modes):
            results.append(_kd_aligner(ctx).compute_batch([fixed], [moving], x0_b, want_stats=True))
    xset(ctx, kd_lds_nodes=1536, kd_modes=1)
    res = results[0]
    for other in results[1:]:
        assert np.array_equal(res.pose, other.pose) and np.array_equal(res.information, other.information) and np.array_equal(res.status, other.status)
    osp = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5)
    for i in range(0, 12, 3):
        s = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        rt = po.align(po.aligner_params(20, device_order=True), [osp], [wl.map_points], [s], x0_b[i])
        _assert_bitwise_equal_to_device_order_oracle(res, i, rt, ("kd role B", i))
    dt = np.abs(res.pose - xt_b)
    assert np.all(res.status == 0) and dt[:, :2].max() < 5e-3 and dt[:, 2].max() < 2e-3
    # role A, ragged scans, Cauchy, non-default leaf parameters
    wl2 = synth.make_workload(6, 30000, seed=7)
    fixed2 = api.CloudSet(ctx, wl2.scan_points, wl2.scan_offsets); moving2 = api.CloudSet(ctx, wl2.map_points)
    al = _kd_aligner(ctx, md=0.3, leaf_range=0.03, leaf_points=10, robustifier=api.RobustifierCauchy(0.05))
    res2 = al.compute_batch([fixed2], [moving2], wl2.x0, want_stats=True)
    if xset(ctx, kd_modes=0):
        try:
            shared = al.compute_batch([fixed2], [moving2], wl2.x0, want_stats=True)
        finally:
            xset(ctx, kd_modes=1)
        assert np.array_equal(res2.pose, shared.pose) and np.array_equal(res2.information, shared.information) and np.array_equal(res2.stats, shared.stats)
    osp2 = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3, kd_max_leaf_range=0.03, kd_min_leaf_points=10, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.05)
    for i in (0, 3, 5):
        s = wl2.scan_points[wl2.scan_offsets[i]:wl2.scan_offsets[i + 1]]
        rt = po.align(po.aligner_params(20, device_order=True), [osp2], [s], [wl2.map_points], wl2.x0[i])
        _assert_bitwise_equal_to_device_order_oracle(res2, i, rt, ("kd role A", i))


def test_mixed_finders_with_a_kdtree_slice(ctx, po, small_workload):
    """projective + KD-tree slices sharing one pose: the k_align<true, true, true, true> instantiation."""
    wl = small_workload
    s = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector()), min_num_correspondences=10))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(_kd_finder(ctx, 0.3), min_num_correspondences=10))
    al.setFixed({"points": s}); al.setMoving({"points": wl.map_points}); al.setMovingInFixed(wl.x0[0])
    assert al.compute() == 0
    osl = [po.slice_params(), po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3)]
    r = po.align(po.aligner_params(10, device_order=True), osl, [s, s], [wl.map_points, wl.map_points], wl.x0[0])
    assert np.array_equal(al.movingInFixed(), r["pose"]) and np.array_equal(al.informationMatrix(), r["H"])


def test_termination_chi_epsilon_all_aligner_paths(ctx, po, small_workload):
    """lsm2d_aligner_params.termination_chi_epsilon (the device-side counterpart of the aligner's termination_criteria): the loop stops
    where the oracle's stops, bit for bit, in the batch kernel, the latency kernel and the split path; 0 keeps max_iterations."""
    wl = synth.make_workload(6, 20000, seed=3, map_noise=0.01, scan_noise=0.01)      # noisy data: the criterion fires at 3 ... 9 iterations, or never
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    n = len(wl.x0)
    for eps in (1e-4, 1e-3, 1e-1):
        outs = []
        for path in (1, 3, 2):
            ctx.set_option("align_path", path)
            al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10, termination_chi_epsilon=eps)
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector()), min_num_correspondences=10))
            res = al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
            assert ctx.get_option("last_align_path") == path
            outs.append(res)
        ctx.set_option("align_path", 0)
        for i in range(n):
            s = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
            rt = po.align(po.aligner_params(20, device_order=True, termination_chi_epsilon=eps), [po.slice_params()], [s], [wl.map_points], wl.x0[i])
            assert 2 <= rt["iterations"] <= 20 and (eps < 1e-3 or rt["iterations"] < 20)
            for res in outs:
                _assert_bitwise_equal_to_device_order_oracle(res, i, rt, ("eps", eps, i))
    with pytest.raises(api.Lsm2dError):
        _neg_eps(ctx, fixed, moving, wl)
    # the two options round 3 refused run on the device since round 4 (test_pair_digest_inlier_only_runs_and_kept_correspondences_all_paths holds them
    # to the oracle); on a slice without robustifier the second loop is five more regular iterations
    al = api.MultiAligner2D(ctx, max_iterations=5)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector())))
    al.param_enable_inlier_only_runs = True; al.param_keep_only_inlier_correspondences = True
    r10 = al.compute_batch([fixed], [moving], wl.x0)
    al2 = api.MultiAligner2D(ctx, max_iterations=10)
    al2.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector())))
    r10b = al2.compute_batch([fixed], [moving], wl.x0)
    assert np.all(r10.iterations == 10) and np.array_equal(r10.pose, r10b.pose) and np.array_equal(r10.information, r10b.information)


def test_start_poses_that_are_not_numbers_fail_their_alignment_not_the_call(ctx, small_workload):
    """NaN / infinite start poses (a caller's bug, a diverged odometry) must cost THEIR alignments a failure status -- every workgroup still reports, the call succeeds,
    the alignments next to them are untouched -- on the batch kernel (with its placement estimate) and on the latency kernel."""
    wl = small_workload
    al = _aligner(ctx)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    for n in (len(wl.x0), 300):
        fi = (np.arange(n, dtype=np.int32) % len(wl.x0)).reshape(1, n)
        x = wl.x0[fi[0]].astype(np.float32).copy()
        good = al.compute_batch([fixed], [moving], x, fixed_index=fi)
        bad = x.copy(); bad[1, 0] = np.nan; bad[3, 2] = np.inf; bad[4, :] = np.nan
        r = al.compute_batch([fixed], [moving], bad, fixed_index=fi)
        ok = np.ones(n, bool); ok[[1, 3, 4]] = False
        assert (r.status[[1, 3, 4]] != 0).all(), r.status[:6]
        assert np.array_equal(r.pose[ok], good.pose[ok]) and np.array_equal(r.status[ok], good.status[ok])
    fixed.close(); moving.close()


def test_point_query_finders_against_the_reference_arithmetic_mode(ctx, po):
    """test_hip_path_against_the_reference_arithmetic_mode for the other finders: the exact grid NN, the reference's own KD-tree and the
    distance map on the device against the oracle in the REFERENCE'S OWN ARITHMETIC (`_r`: libm, no FMA, Eigen's association, sums pair
    after pair).  Poses within the north_star tolerance of 1e-4 m / 1e-4 rad; the fraction of first-iteration pairs that differ is
    reported per finder (PARITY.md section 5)."""
    wl = synth.make_workload(8, 100000, seed=12)
    x0_b = synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32)
    fixed = api.CloudSet(ctx, wl.map_points); moving = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    finders = (("exact NN", api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5), po.slice_params(finder=po.FINDER_NN, max_distance=0.5)),
               ("KD-tree", api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.5, search="kdtree"), po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5)),
               ("distance map", api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05), po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.05)))
    report = []
    for name, finder, osp in finders:
        al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10))
        res = al.compute_batch([fixed], [moving], x0_b)
        worst = np.zeros(2); differing = []
        for i in range(len(x0_b)):
            sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
            ref = po.align(po.aligner_params(20), [osp], [wl.map_points], [sc], x0_b[i], double="ref")
            assert ref["status"] == 0 and res.status[i] == 0, (name, i)
            d = np.abs(res.pose[i] - ref["pose"]); worst = np.maximum(worst, [d[:2].max(), d[2]])
            finder.setFixed(fixed, 0); finder.setMoving(moving, i); finder.setLocalMapInSensor(x0_b[i])
            got = {tuple(p) for p in finder.compute().tolist()}
            want = {tuple(p) for p in po.find(osp, wl.map_points, sc, x0_b[i], double="ref").tolist()}
            differing.append(len(got ^ want) / max(1, len(want)))
        assert worst[0] < POSE_TOL_M and worst[1] < POSE_TOL_RAD, (name, worst)
        report.append("%s: max pose delta %.1e m / %.1e rad, pairs differing at x0 %.3f %% (mean)" % (name, worst[0], worst[1], 100 * float(np.mean(differing))))
    print("HIP vs reference arithmetic, point-query finders, role B, 100k map: " + "; ".join(report))


def test_pair_digest_inlier_only_runs_and_kept_correspondences_all_paths(ctx, po):
    """Round 4: (i) every iteration's statistics carry the order-independent digest of its correspondence SET -- equal to the oracle's in every
    finder kind, role and aligner path (and lsm2d_linearize's to the host-side hash of the pairs it was given); (ii) MultiAligner2D's
    enable_inlier_only_runs runs the second loop on the device, bit for bit the device-order mirror's, in the three aligner paths and with a
    point-query finder; (iii) lsm2d_align_batch_pairs hands back what the reference leaves in slice->correspondences(): the last iteration's
    pairs, exactly the oracle's, only the inliers under keep_only_inlier_correspondences (MULTI.json:606-610; apps/visual_test_aligner_2d.cpp:129-143)."""
    world = synth.make_world(4)
    m = synth.make_map(world, 30000, noise_sigma=0.0, seed=2)
    robots = synth.sample_poses(world, 3, seed=8)
    pts, offs = synth.make_scans(world, robots, n_beams=721, noise_sigma=0.02, seed=5)      # range noise: outliers under a tight kernel, to the end
    x0 = synth.invert_poses(synth.compose_poses(robots, np.tile([[0.12, -0.08, 0.04]], (3, 1)))).astype(np.float32)
    scans = [pts[offs[i]:offs[i + 1]] for i in range(3)]
    tau = 5e-4

    def run(al, path, *a, **kw):
        ctx.set_option("align_path", path)
        try:
            return al.compute_batch(*a, **kw)
        finally:
            ctx.set_option("align_path", 0)

    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 25.0)
    fx = api.CloudSet(ctx, pts, offs); mv = api.CloudSet(ctx, m)
    osp = po.slice_params(canvas_cols=721, range_max=25.0, robustifier=po.ROBUST_CAUCHY, chi_threshold=tau, min_num_correspondences=5)
    for eps in (0.0, 2e-2):
        for inl, keep in ((False, False), (True, False), (True, True), (False, True)):
            al = api.MultiAligner2D(ctx, max_iterations=7, min_num_inliers=10, termination_chi_epsilon=eps)
            al.param_enable_inlier_only_runs = inl; al.param_keep_only_inlier_correspondences = keep
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8),
                                                                              robustifier=api.RobustifierCauchy(tau), min_num_correspondences=5))
            oap = po.aligner_params(7, device_order=True, termination_chi_epsilon=eps, enable_inlier_only_runs=inl, keep_only_inlier_correspondences=keep)
            want = [po.align(oap, [osp], [scans[i]], [m], x0[i], want_pairs=True) for i in range(3)]
            for path in (1, 2, 3):
                r = run(al, path, [fx], [mv], x0, want_stats=True, want_pairs=True)
                assert ctx.get_option("last_align_path") == path
                assert r.stats.shape[1] == (14 if inl else 7)
                for i in range(3):
                    _assert_bitwise_equal_to_device_order_oracle(r, i, want[i], ("path %d inl %d keep %d eps %g" % (path, inl, keep, eps), i))
                    assert np.array_equal(r.pairs[i][0], want[i]["pairs"][0]), (path, inl, keep, i, len(r.pairs[i][0]), len(want[i]["pairs"][0]))
                    last = r.stats[i][r.iterations[i] - 1]
                    if keep:
                        assert len(r.pairs[i][0]) == last["n_inliers"] < last["n_correspondences"]
                    else:      # the unfiltered vector IS the last iteration's correspondence set: its digest, formed on the host
                        assert len(r.pairs[i][0]) == last["n_correspondences"]
                        assert po.pair_digest(r.pairs[i][0]) == int(api.pair_digests(r.stats[i][r.iterations[i] - 1: r.iterations[i]])[0])
            if inl and eps == 0.0:
                assert all(w["iterations"] == 14 for w in want)
    # two slices with sensor offsets + prior, point-query finders in both roles: digests (inside the bitwise check) and the pairs that come back
    S0 = np.float32([0.2, 0.1, 0.1])
    sc0 = synth.make_scans(world, synth.compose_poses(robots, np.tile(S0[None, :].astype(np.float64), (3, 1))), n_beams=541, noise_sigma=0.01, seed=9)
    al2 = api.MultiAligner2D(ctx, max_iterations=5, min_num_inliers=5)
    al2.param_enable_inlier_only_runs = True; al2.param_keep_only_inlier_correspondences = True
    f_nn = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4, normal_cos=0.7)
    al2.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=S0,
                                                                                 robustifier=api.RobustifierCauchy(2e-3), min_num_correspondences=5))
    al2.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(f_nn, robustifier=api.RobustifierCauchy(1e-3), min_num_correspondences=5))
    fx2 = [api.CloudSet(ctx, sc0[0], sc0[1]), fx]; mv2 = [mv, mv]
    pri = [(x0[i], np.eye(3, dtype=np.float32) * 10.0) for i in range(3)]
    r2 = al2.compute_batch(fx2, mv2, x0, priors=pri, want_stats=True, want_pairs=True)
    osl = [_oracle_slice(po, s_.slice_params()) for s_ in al2.param_slice_processors]
    for i in range(3):
        w = po.align(po.aligner_params(5, min_num_inliers=5, device_order=True, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True,
                                       prior_z=pri[i][0], prior_omega=pri[i][1]), osl, [sc0[0][sc0[1][i]:sc0[1][i + 1]], scans[i]], [m, m], x0[i], want_pairs=True)
        _assert_bitwise_equal_to_device_order_oracle(r2, i, w, ("two slices", i))
        for s_ in range(2):
            assert np.array_equal(r2.pairs[i][s_], w["pairs"][s_]), (i, s_, len(r2.pairs[i][s_]), len(w["pairs"][s_]))
    # every point-query finder, both roles, with the second loop: bitwise incl. the digests
    small = m[::6].copy()
    for kind in ("exact", "kdtree", "distmap"):
        for role in ("A", "B"):
            f = (api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.4, resolution=0.1, normal_cos=0.7) if kind == "distmap"
                 else api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4, normal_cos=0.7, search=kind))
            al3 = api.MultiAligner2D(ctx, max_iterations=4, min_num_inliers=5); al3.param_enable_inlier_only_runs = True
            al3.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(f, robustifier=api.RobustifierCauchy(2e-3), min_num_correspondences=5))
            o3 = _oracle_slice(po, al3.param_slice_processors[0].slice_params())
            if role == "A":
                r3 = al3.compute_batch([fx], [api.CloudSet(ctx, small)], x0, want_stats=True, want_pairs=True)
                w3 = [po.align(po.aligner_params(4, min_num_inliers=5, device_order=True, enable_inlier_only_runs=True), [o3], [scans[i]], [small], x0[i], want_pairs=True) for i in range(3)]
            else:
                xb = synth.invert_poses(x0.astype(np.float64)).astype(np.float32)
                r3 = al3.compute_batch([api.CloudSet(ctx, small)], [fx], xb, want_stats=True, want_pairs=True)
                w3 = [po.align(po.aligner_params(4, min_num_inliers=5, device_order=True, enable_inlier_only_runs=True), [o3], [small], [scans[i]], xb[i], want_pairs=True) for i in range(3)]
            for i in range(3):
                _assert_bitwise_equal_to_device_order_oracle(r3, i, w3[i], (kind, role, i))
                assert np.array_equal(r3.pairs[i][0], w3[i]["pairs"][0]), (kind, role, i)
    # lsm2d_linearize: the digest of the pairs it was handed (slice 0), through the kernels' hash
    pr0 = po.find(po.slice_params(canvas_cols=721, range_max=25.0), scans[0], m, x0[0])
    _, _, st = api.linearize(ctx, al.param_slice_processors[0].slice_params(), scans[0], m, pr0, x0[0])
    assert st.pair_digest == po.pair_digest(pr0) and st.n_correspondences == len(pr0)
    # capacity and argument checks of the pairs call
    lib = ctx._lib
    import ctypes as C
    from srrg2_laser_slam_2d_amd import _capi
    sp = (_capi.SliceParams * 1)(al.param_slice_processors[0].slice_params())
    b = _capi.Batch(); b.n_alignments, b.n_slices = 1, 1; b.slices = sp
    h_f = (C.c_void_p * 1)(fx.handle.value); h_m = (C.c_void_p * 1)(mv.handle.value)
    b.fixed = C.cast(h_f, C.POINTER(C.c_void_p)); b.moving = C.cast(h_m, C.POINTER(C.c_void_p))
    idx = np.zeros(1, np.int32); b.fixed_index = idx.ctypes.data_as(C.POINTER(C.c_int32))
    xx = x0[:1].copy(); b.init_pose = xx.ctypes.data_as(C.POINTER(C.c_float))
    ap = _capi.AlignerParams(3, 5, 0.0, 0.0, 0, 0)
    pose = np.empty(3, np.float32); status = np.empty(1, np.int32); buf = np.empty((721, 2), np.int32); cnt = np.zeros(1, np.int32)
    rc = lib.lsm2d_align_batch_pairs(ctx.handle, C.byref(ap), C.byref(b), pose.ctypes.data_as(C.c_void_p), None, status.ctypes.data_as(C.c_void_p), None, None,
                                     buf.ctypes.data_as(C.c_void_p), 720, cnt.ctypes.data_as(C.c_void_p))
    assert rc == _capi.CAPACITY_EXCEEDED
    rc = lib.lsm2d_align_batch_pairs(ctx.handle, C.byref(ap), C.byref(b), pose.ctypes.data_as(C.c_void_p), None, status.ctypes.data_as(C.c_void_p), None, None,
                                     buf.ctypes.data_as(C.c_void_p), 721, cnt.ctypes.data_as(C.c_void_p))
    assert rc == 0 and 0 < cnt[0] <= 721 and lib.lsm2d_stats_capacity(C.byref(ap)) == 3
    ap2 = _capi.AlignerParams(3, 5, 0.0, 0.0, 1, 0); assert lib.lsm2d_stats_capacity(C.byref(ap2)) == 6
