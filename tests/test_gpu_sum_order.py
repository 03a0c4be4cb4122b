"""GPU tests of the option "sum_order" 1 (round 6): H, b and the chi^2 statistics added PAIR AFTER PAIR in the reference's order
(octave/solver/nicp_post.m:69-90; registration/correspondence_finder_projective_2d.cpp:55-74: ascending column;
registration/correspondence_finder_kd_tree_2d.cpp:12-27: ascending moving index).

The bar is the strongest the oracle offers: with the option on, the HIP path equals the SEQUENTIAL fp32 oracle -- lsmo_align_f / lsmo_linearize_f with
device_order = 0, the restatement that was written from the reference's files, not after the device -- BIT FOR BIT: status, iteration count, pose,
information matrix, every iteration's counts, chi^2 sums and pair digest.  No tolerance appears in this file."""
import math

import numpy as np
import pytest

from srrg2_laser_slam_2d_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def seq_ctx(ctx):
    ctx.set_option("sum_order", 1)
    try:
        yield ctx
    finally:
        ctx.set_option("sum_order", 0)
        ctx.set_option("align_path", 0)


def _projector(cols=1081, rmin=0.3, rmax=30.0):
    return api.PointNormal2fProjectorPolar(cols, -math.pi, math.pi, rmin, rmax)


def assert_bitwise(res, i, ro, tag):
    """res: BatchResult of the device; ro: pyoracle.align(...) of the sequential fp32 oracle"""
    assert int(res.status[i]) == ro["status"] and int(res.iterations[i]) == ro["iterations"], (tag, int(res.status[i]), ro["status"], int(res.iterations[i]), ro["iterations"])
    assert np.array_equal(res.pose[i], ro["pose"]), (tag, "pose", res.pose[i].tolist(), ro["pose"].tolist())
    assert np.array_equal(res.information[i], ro["H"]), (tag, "H", res.information[i].tolist(), ro["H"].tolist())
    if res.stats is not None:
        for k in range(ro["iterations"]):
            g, o = res.stats[i][k], ro["stats"][k]
            assert (int(g["n_correspondences"]), int(g["n_inliers"]), int(g["n_outliers"])) == (o.n_corr, o.n_in, o.n_out), (tag, "counts", k)
            assert np.float32(g["chi_inliers"]) == np.float32(o.chi_in) and np.float32(g["chi_outliers"]) == np.float32(o.chi_out), (tag, "chi", k)
            assert (int(g["pair_digest_hi"]) << 32 | int(g["pair_digest_lo"])) == (o.pair_digest_hi << 32 | o.pair_digest_lo), (tag, "pair digest", k)


def test_factor_pair_after_pair_is_the_sequential_oracle_bit_for_bit(seq_ctx, po, small_workload):
    """lsm2d_linearize: a few hundred pairs (one trip of the workgroup), thousands (many trips), none; with and without Cauchy."""
    ctx, wl = seq_ctx, small_workload
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    for osp_find in (po.slice_params(), po.slice_params(finder=po.FINDER_NN, max_distance=0.3)):
        corr = po.find(osp_find, f, wl.map_points, wl.x0[0])
        assert len(corr) > 300
        for robust in (api.ROBUST_NONE, api.ROBUST_CAUCHY):
            H, b, st = api.linearize(ctx, api.make_slice_params(robustifier=robust, chi_threshold=0.002), f, wl.map_points, corr, wl.x0[0])
            oH, ob, ost = po.linearize(po.slice_params(robustifier=robust, chi_threshold=0.002), f, wl.map_points, corr, wl.x0[0])
            assert np.array_equal(H, oH) and np.array_equal(b, ob), (len(corr), robust, (H - oH).tolist())
            assert (st.n_correspondences, st.n_inliers, st.n_outliers) == (len(corr), ost.n_in, ost.n_out)
            assert np.float32(st.chi_inliers) == np.float32(ost.chi_in) and np.float32(st.chi_outliers) == np.float32(ost.chi_out)
            assert (st.pair_digest_hi, st.pair_digest_lo) == (ost.pair_digest_hi, ost.pair_digest_lo)
            if robust == api.ROBUST_CAUCHY:
                assert ost.n_out > 0      # the outlier branch (the logarithm) took part
    H, b, st = api.linearize(ctx, api.make_slice_params(), f, wl.map_points, np.zeros((0, 2), np.int32), wl.x0[0])
    assert np.all(H == 0) and np.all(b == 0) and st.n_correspondences == 0


def test_projective_aligner_culled_stream_and_small_clouds(seq_ctx, po):
    """The headline shape (scans against a 100k-point map: k_align_seq<1,0,0,0,5>, the culled stream) and a small moving cloud (mode 0: no lane copy), both with
    and without Cauchy, batches through an index array; and the order must not depend on where an alignment runs (culling / placement on or off)."""
    ctx = seq_ctx
    wl = synth.make_workload(12, 100000, seed=11, map_noise=0.004, scan_noise=0.004)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    for rb, tau in ((None, 0.0), (api.RobustifierCauchy(0.01), 0.01)):
        al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector()), min_num_correspondences=10, robustifier=rb))
        n = 300      # > 256: travels by copies, placed by estimated work
        fi = (np.arange(n, dtype=np.int32) % 12).reshape(1, n)
        x0 = wl.x0[fi[0]].astype(np.float32).copy(); x0[:, 0] += np.linspace(-0.01, 0.01, n, dtype=np.float32)
        res = al.compute_batch([fixed], [moving], x0, fixed_index=fi, want_stats=True)
        assert ctx.get_option("last_align_path") == 1
        osp = po.slice_params(robustifier=po.ROBUST_CAUCHY if rb else po.ROBUST_NONE, chi_threshold=tau if rb else 0.05)
        for i in (0, 1, 7, 150, 299):
            c = int(fi[0, i])
            ro = po.align(po.aligner_params(20), [osp], [wl.scan_points[wl.scan_offsets[c]:wl.scan_offsets[c + 1]]], [wl.map_points], x0[i])
            assert_bitwise(res, i, ro, ("culled", bool(rb), i))
        ctx.set_option("cull", 0); ctx.set_option("balance", 0)
        try:
            res2 = al.compute_batch([fixed], [moving], x0, fixed_index=fi, want_stats=True)
        finally:
            ctx.set_option("cull", 1); ctx.set_option("balance", 1)
        assert np.array_equal(res.pose, res2.pose) and np.array_equal(res.information, res2.information) and np.array_equal(res.stats, res2.stats)
    # (round 6, late) a PACKED batch in the reference's order: 1040 alignments in one round of 1024 workgroups, the lightest two to a workgroup (k_align_seq_two)
    al = api.MultiAligner2D(ctx, max_iterations=6, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector()), min_num_correspondences=10))
    n = 1040
    fi = (np.arange(n, dtype=np.int32) % 12).reshape(1, n)
    x0 = wl.x0[fi[0]].astype(np.float32).copy(); x0[:, 1] += np.linspace(-0.01, 0.01, n, dtype=np.float32)
    res = al.compute_batch([fixed], [moving], x0, fixed_index=fi, want_stats=True)
    assert ctx.get_option("last_align_width") == 1024 and ctx.get_option("last_align_path") == 1
    ctx.set_option("align_width", 512)
    try:
        res2 = al.compute_batch([fixed], [moving], x0, fixed_index=fi, want_stats=True)
        assert ctx.get_option("last_align_width") == 512
    finally:
        ctx.set_option("align_width", 0)
    assert np.array_equal(res.pose, res2.pose) and np.array_equal(res.information, res2.information) and np.array_equal(res.stats, res2.stats) and np.array_equal(res.status, res2.status)
    for i in (0, 517, 1039):
        c = int(fi[0, i])
        ro = po.align(po.aligner_params(6), [po.slice_params()], [wl.scan_points[wl.scan_offsets[c]:wl.scan_offsets[c + 1]]], [wl.map_points], x0[i])
        assert_bitwise(res, i, ro, ("packed", i))
    # a small moving cloud, canvas smaller and larger than one trip of the workgroup
    small = synth.make_workload(4, 700, seed=12)
    for cols in (300, 512, 513, 1500):
        al = api.MultiAligner2D(ctx, max_iterations=8, min_num_inliers=5)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector(cols)), min_num_correspondences=5))
        fx = api.CloudSet(ctx, small.scan_points, small.scan_offsets); mv = api.CloudSet(ctx, small.map_points)
        res = al.compute_batch([fx], [mv], small.x0, want_stats=True)
        assert ctx.get_option("last_align_path") == 1      # (the latency kernel keeps the tree order: not taken with the option on)
        for i in range(4):
            ro = po.align(po.aligner_params(8, min_num_inliers=5), [po.slice_params(canvas_cols=cols, min_num_correspondences=5)],
                          [small.scan_points[small.scan_offsets[i]:small.scan_offsets[i + 1]]], [small.map_points], small.x0[i])
            assert_bitwise(res, i, ro, ("small", cols, i))


def test_point_query_finders_both_roles(seq_ctx, po):
    """Exact NN (grid), the reference's KD-tree and the distance map, role A (the tracker's wiring: every map point a query -- tens of trips, tiles culled) and
    role B (BASELINE's wording: scans query the map; the NN finder's cooperative search, four lanes per query)."""
    ctx = seq_ctx
    wl = synth.make_workload(6, 30000, seed=21, map_noise=0.003, scan_noise=0.003)
    x0_b = synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32)
    scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); themap = api.CloudSet(ctx, wl.map_points)
    finders = [
        ("nn", lambda: api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3), po.slice_params(finder=po.FINDER_NN, max_distance=0.3)),
        ("kdtree", lambda: api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, search="kdtree"), po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3)),
        ("distmap", lambda: api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.3, resolution=0.05), po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.3, resolution=0.05)),
    ]
    for name, make, osp in finders:
        for cauchy in (False, True):
            osp.robustifier = po.ROBUST_CAUCHY if cauchy else po.ROBUST_NONE; osp.chi_threshold = 0.004; osp.min_num_correspondences = 10
            al = api.MultiAligner2D(ctx, max_iterations=6, min_num_inliers=10)
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(make(), min_num_correspondences=10, robustifier=api.RobustifierCauchy(0.004) if cauchy else None))
            ra = al.compute_batch([scans], [themap], wl.x0, want_stats=True)          # role A
            rb = al.compute_batch([themap], [scans], x0_b, want_stats=True)           # role B
            for i in (0, 3, 5):
                s = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
                assert_bitwise(ra, i, po.align(po.aligner_params(6), [osp], [s], [wl.map_points], wl.x0[i]), (name, cauchy, "role A", i))
                assert_bitwise(rb, i, po.align(po.aligner_params(6), [osp], [wl.map_points], [s], x0_b[i]), (name, cauchy, "role B", i))


def test_two_and_three_slices_mixed_finders_sensor_offsets_prior_and_options(seq_ctx, po):
    """The tracker's structure: several slices (their totals added in slice order), sensor offsets, an odometry prior, a slice below min_num_correspondences,
    termination_chi_epsilon, the inlier-only runs."""
    ctx = seq_ctx
    world = synth.make_world(7)
    m = synth.make_map(world, 30000, noise_sigma=0.003, seed=33)
    robots = synth.sample_poses(world, 3, seed=17)
    rng = np.random.default_rng(6)
    guess = synth.compose_poses(robots, rng.uniform(-0.04, 0.04, (3, 3)))
    x0 = synth.invert_poses(guess).astype(np.float32)
    S = [np.float32([0.2, -0.1, 0.5]), np.zeros(3, np.float32), np.float32([-0.25, 0.05, -2.9])]
    mv = api.CloudSet(ctx, m)
    for variant in range(4):
        ns = 2 if variant == 0 else 3
        kw = [dict(), dict(termination_chi_epsilon=1e-3), dict(enable_inlier_only_runs=True), dict()][variant]
        al = api.MultiAligner2D(ctx, max_iterations=9, min_num_inliers=10, termination_chi_epsilon=kw.get("termination_chi_epsilon", 0.0))
        al.param_enable_inlier_only_runs = bool(kw.get("enable_inlier_only_runs", False))
        oslices, fixed_sets, scans = [], [], []
        for s in range(ns):
            mixed = variant == 3
            if mixed and s == 1:
                f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4, normal_cos=0.7)
            elif mixed and s == 2:
                f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.4, resolution=0.08, normal_cos=0.7)
            else:
                f = api.CorrespondenceFinderProjective2f(ctx, _projector(700 + 190 * s, rmax=20.0), 0.6, 0.7)
            rob = api.RobustifierCauchy(0.02) if s != 1 else None
            mc = 5 if s < 2 else 100000 if variant == 1 else 5      # variant 1: the third slice never has enough pairs and is skipped
            sl = (api.AlignerSliceProcessorLaser2DWithSensor(f, sensor_in_robot=S[s], robustifier=rob, min_num_correspondences=mc) if S[s].any()
                  else api.AlignerSliceProcessorLaser2D(f, robustifier=rob, min_num_correspondences=mc))
            al.param_slice_processors.append(sl)
            pts, offs = synth.make_scans(world, synth.compose_poses(robots, np.tile(S[s][None, :].astype(np.float64), (3, 1))), n_beams=500 + 150 * s, noise_sigma=0.003, seed=40 + s)
            fixed_sets.append(api.CloudSet(ctx, pts, offs)); scans.append((pts, offs))
            sp = sl.slice_params()
            oslices.append(po.slice_params(finder=sp.finder, canvas_cols=sp.projector.canvas_cols, angle_min=sp.projector.angle_min, angle_max=sp.projector.angle_max,
                                           range_min=sp.projector.range_min, range_max=sp.projector.range_max, col_offset=sp.projector.col_offset,
                                           point_distance=sp.point_distance, normal_cos=sp.normal_cos, max_distance=sp.max_distance, resolution=sp.resolution,
                                           robustifier=sp.robustifier, chi_threshold=sp.chi_threshold, min_num_correspondences=sp.min_num_correspondences,
                                           sensor_in_robot=tuple(sp.sensor_in_robot), kd_max_leaf_range=sp.kd_max_leaf_range, kd_min_leaf_points=sp.kd_min_leaf_points))
        pri = [(x0[i].copy(), np.diag([30.0, 20.0, 50.0]).astype(np.float32)) for i in range(3)]
        paths = (1, 2) if variant != 3 else (1,)      # the split path takes projective slices only
        for path in paths:
            ctx.set_option("align_path", path)
            res = al.compute_batch(fixed_sets, [mv] * ns, x0, priors=pri, want_stats=True)
            assert ctx.get_option("last_align_path") == path
            for i in range(3):
                sc = [p[o[i]:o[i + 1]] for p, o in scans]
                ro = po.align(po.aligner_params(9, prior_z=pri[i][0], prior_omega=pri[i][1], **kw), oslices, sc, [m] * ns, x0[i])
                assert_bitwise(res, i, ro, ("variant", variant, "path", path, i))
        ctx.set_option("align_path", 0)


def test_asynchronous_begin_wait_and_single_alignment_calls_keep_the_order(seq_ctx, po, small_workload):
    """n = 1 (results through pinned memory, start pose in the kernel arguments) and the begin / wait form."""
    ctx, wl = seq_ctx, small_workload
    al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector()), min_num_correspondences=10))
    f = wl.scan_points[wl.scan_offsets[2]:wl.scan_offsets[3]]
    ro = po.align(po.aligner_params(20), [po.slice_params()], [f], [wl.map_points], wl.x0[2])
    al.setFixed({"points": f}); al.setMoving({"points": wl.map_points}); al.setMovingInFixed(wl.x0[2])
    assert al.compute() == ro["status"]
    assert np.array_equal(al.movingInFixed(), ro["pose"]) and np.array_equal(al.informationMatrix(), ro["H"])
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    prep = al.prepare_batch([fixed], [moving], wl.x0, want_stats=True)
    prep.begin(); res = prep.wait(copy=True)
    for i in range(len(wl.x0)):
        ro = po.align(po.aligner_params(20), [po.slice_params()], [wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]], [wl.map_points], wl.x0[i])
        assert_bitwise(res, i, ro, ("begin / wait", i))
    # and the default order is still the device-order mirror's: switching the option off restores it
    ctx.set_option("sum_order", 0)
    res0 = al.compute_batch([fixed], [moving], wl.x0)
    rt = po.align(po.aligner_params(20, device_order=True), [po.slice_params()], [wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]], [wl.map_points], wl.x0[0])
    assert np.array_equal(res0.pose[0], rt["pose"]) and np.array_equal(res0.information[0], rt["H"])
