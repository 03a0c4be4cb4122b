"""GPU parity tests, row a10, launch forms: the split path, the latency kernel, culling and placement, prepared batches, batches in flight -- every form the same bits.

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


def test_device_resident_input(ctx, small_workload):
    import torch
    wl = small_workload
    t = torch.from_numpy(wl.map_points).cuda()
    a = api.CloudSet(ctx, t); b = api.CloudSet(ctx, wl.map_points)
    pr = _projector()
    sa = pr.compute(ctx, a, wl.x0[0]); sb = pr.compute(ctx, b, wl.x0[0])
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])


def test_split_path_is_bit_identical_to_fused_path(ctx, po):
    """The many-workgroups-per-alignment path (k_split_project + k_split_finish) against the one-workgroup path (k_align):
    same z-buffer winners (u64 min is order independent), same reduction order -> bitwise equal poses, H, statistics."""
    wl = synth.make_workload(5, 200000, seed=21)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    x0 = wl.x0.copy(); x0[3] += np.float32([70, 70, 0])                     # one hopeless candidate: NotEnoughCorrespondences
    def run(path, al, *args, **kw):
        ctx.set_option("align_path", path)
        try:
            return al.compute_batch(*args, **kw)
        finally:
            ctx.set_option("align_path", 0)
    al = _aligner(ctx)
    a = run(1, al, [fixed], [moving], x0, want_stats=True); b = run(2, al, [fixed], [moving], x0, want_stats=True)
    for k in ("pose", "information", "status", "iterations"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    for i in range(5):
        assert np.array_equal(a.stats[i][: a.iterations[i]], b.stats[i][: b.iterations[i]])
    assert a.status[3] == 1 and a.iterations[3] == 1 and np.all(a.status[[0, 1, 2, 4]] == 0)
    # automatic choice: a single alignment against a big map takes the split path and matches the oracle
    c = al.compute_batch([api.CloudSet(ctx, wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]])], [moving], x0[:1])
    assert np.array_equal(c.pose[0], a.pose[0])
    r = po.align(po.aligner_params(20), [po.slice_params()], [wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]], [wl.map_points], x0[0])
    d = np.abs(c.pose[0] - r["pose"]); assert d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    # NotEnoughInliers and SingularH through the split path
    al2 = _aligner(ctx); al2.param_min_num_inliers = 100000
    assert np.all(run(2, al2, [fixed], [moving], wl.x0).status == 2)
    wall = np.stack([np.linspace(-3, 3, 400), np.full(400, 2.0), np.zeros(400), -np.ones(400)], 1).astype(np.float32)
    al3 = _aligner(ctx, 360); al3.param_slice_processors[0].param_min_num_correspondences = 0
    assert run(2, al3, [wall], [wall], np.zeros((1, 3), np.float32)).status[0] == 3
    # multi-slice with extrinsics, Cauchy and prior
    world = synth.make_world(5); m = synth.make_map(world, 80000)
    robot = synth.sample_poses(world, 2, seed=11)
    S0, S1 = np.array([0.2, 0.1, 0.1]), np.array([-0.3, 0.0, math.pi])
    sc = [synth.make_scans(world, synth.compose_poses(robot, np.tile(S, (2, 1))), n_beams=721) for S in (S0, S1)]
    xg = synth.invert_poses(synth.compose_poses(robot, np.tile([[0.04, -0.03, 0.03]], (2, 1)))).astype(np.float32)
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    alm = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    alm.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.9), sensor_in_robot=S0,
                                                                                 robustifier=api.RobustifierCauchy(0.01), min_num_correspondences=5))
    alm.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=S1,
                                                                                 min_num_correspondences=5))
    fx = [api.CloudSet(ctx, p, o) for p, o in sc]; mv = [api.CloudSet(ctx, m)] * 2
    pri = [(xg[i], np.eye(3, dtype=np.float32) * 20.0) for i in range(2)]
    f1 = run(1, alm, fx, mv, xg, priors=pri, want_stats=True); f2 = run(2, alm, fx, mv, xg, priors=pri, want_stats=True)
    assert np.array_equal(f1.pose, f2.pose) and np.array_equal(f1.information, f2.information) and np.array_equal(f1.stats, f2.stats)
    assert np.all(f1.status == 0)


def test_slice_pair_kernel_is_bit_identical_to_one_slice_after_the_other(ctx, po):
    """Two projective slices side by side in one 1024-thread workgroup (k_align_pair, the live tracker's two-scanner aligner)
    against k_align running them one after the other: same thread <-> pair mapping per slice, same gather order, slice totals
    added in slice order -> bitwise equal poses, information matrices, statistics, statuses."""
    def run(path, al, *args, **kw):
        ctx.set_option("align_path", path)
        try:
            r = al.compute_batch(*args, **kw)
            return r, ctx.get_option("last_align_path")
        finally:
            ctx.set_option("align_path", 0)
    def same(a, b):
        for k in ("pose", "information", "status", "iterations"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), k
        for i in range(len(a.status)):
            assert np.array_equal(a.stats[i][: a.iterations[i]], b.stats[i][: b.iterations[i]]), i
    world = synth.make_world(5)
    S0, S1 = np.array([0.2, 0.1, 0.1]), np.array([-0.3, 0.0, math.pi])
    proj0 = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    proj1 = api.PointNormal2fProjectorPolar(541, -math.pi, math.pi, 0.5, 9.0)       # other columns AND range gate: one canvas per slice
    def aligner(min_inliers=10, min_corr=5):
        al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=min_inliers)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
            api.CorrespondenceFinderProjective2f(ctx, proj0, 0.5, 0.9), sensor_in_robot=S0, robustifier=api.RobustifierCauchy(0.01), min_num_correspondences=min_corr))
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
            api.CorrespondenceFinderProjective2f(ctx, proj1, 0.5, 0.8), sensor_in_robot=S1, min_num_correspondences=min_corr))
        return al
    for n_map, n in ((900, 1), (30000, 3), (30000, 300)):      # a clipped-scene sized map (no lane-chunked copy), a streamed one, a big batch (forced)
        m = synth.make_map(world, n_map)
        robot = synth.sample_poses(world, n, seed=11)
        sc = [synth.make_scans(world, synth.compose_poses(robot, np.tile(S, (n, 1))), n_beams=721) for S in (S0, S1)]
        xg = synth.invert_poses(synth.compose_poses(robot, np.tile([[0.04, -0.03, 0.03]], (n, 1)))).astype(np.float32)
        if n >= 3:
            xg[1] += np.float32([70, 70, 0])                   # a hopeless candidate: NotEnoughCorrespondences after one iteration
        fx = [api.CloudSet(ctx, p, o) for p, o in sc]; mv = [api.CloudSet(ctx, m)] * 2
        for pri in (None, [(xg[i], np.eye(3, dtype=np.float32) * 20.0) for i in range(n)]):
            al = aligner()
            (f1, p1), (f3, p3) = run(1, al, fx, mv, xg, priors=pri, want_stats=True), run(3, al, fx, mv, xg, priors=pri, want_stats=True)
            assert p1 == 1 and p3 == 3
            same(f1, f3)
            assert (n_map < 30000 or f1.status[0] == 0) and (n < 3 or (f1.status[1] == 1 and f1.iterations[1] == 1))
            fa, pa = run(0, al, fx, mv, xg, priors=pri, want_stats=True)
            assert pa == (3 if n <= 256 else 1)
            same(f1, fa)
        # the oracle in the kernels' summation order gives the same bits (first alignment, with the prior)
        osl = [_oracle_slice(po, sp.slice_params()) for sp in al.param_slice_processors]
        r = po.align(po.aligner_params(10, prior_z=xg[0], prior_omega=np.eye(3, dtype=np.float32) * 20.0, device_order=True), osl,
                     [sc[0][0][sc[0][1][0]:sc[0][1][1]], sc[1][0][sc[1][1][0]:sc[1][1][1]]], [m, m], xg[0])
        _assert_bitwise_equal_to_device_order_oracle(f3, 0, r, ("pair", n_map, n))
        # NotEnoughInliers; one slice below min_num_correspondences (skipped), both below (NotEnoughCorrespondences)
        al2 = aligner(min_inliers=100000)
        (g1, _), (g3, q3) = run(1, al2, fx, mv, xg, want_stats=True), run(3, al2, fx, mv, xg, want_stats=True)
        assert q3 == 3 and g3.status[0] in (1, 2)
        same(g1, g3)
        al3 = aligner(min_corr=400)                             # the 541-column slice never has that many pairs
        (h1, _), (h3, _) = run(1, al3, fx, mv, xg, want_stats=True), run(3, al3, fx, mv, xg, want_stats=True)
        same(h1, h3)
        al4 = aligner(min_corr=5000)
        (k1, _), (k3, _) = run(1, al4, fx, mv, xg, want_stats=True), run(3, al4, fx, mv, xg, want_stats=True)
        assert np.all(k3.status == 1)
        same(k1, k3)
    # one projective slice: the same kernel with 512 threads (automatic up to 256 alignments); with prior, statuses, statistics
    wl = synth.make_workload(3, 5000, seed=3)
    x0 = wl.x0.copy(); x0[1] += np.float32([70, 70, 0])
    fx, mv = [api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)], [api.CloudSet(ctx, wl.map_points)]
    for pri in (None, [(x0[i], np.eye(3, dtype=np.float32) * 30.0) for i in range(3)]):
        (s1, q1), (s3, q3), (s0, q0) = (run(path, _aligner(ctx), fx, mv, x0, priors=pri, want_stats=True) for path in (1, 3, 0))
        assert (q1, q3, q0) == (1, 3, 3)
        same(s1, s3); same(s1, s0)
        assert s1.status[0] == 0 and s1.status[1] == 1
    # three slices, or another finder: the option falls back to the ordinary kernel
    aln = api.MultiAligner2D(ctx, max_iterations=5, min_num_inliers=10)
    aln.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.8), min_num_correspondences=5))
    r1, p = run(3, aln, fx, mv, wl.x0)
    assert p == 1 and np.all(r1.status == 0)


def test_deferred_upload_is_unpacked_by_whoever_reads_the_set_first(ctx, po, small_workload):
    """lsm2d_cloudset_upload of a scan-sized set only fills the set's pinned buffer; the unpacking is queued by the first reader, or
    done by the aligner kernel in its prologue (single-alignment projective calls: k_align and k_align_pair).  Every reader must see
    the uploaded points, the kernels that unpack must leave them behind for later readers, and results must be those of a set
    created in one go."""
    wl = small_workload
    sc = [wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]] for i in range(3)]
    m = api.CloudSet(ctx, wl.map_points)
    # (1) plain readers: download, size, replaced uploads
    r = api.CloudSet.reserved(ctx, 2048)
    r.upload(sc[0]); assert r.n_points == len(sc[0]) and np.array_equal(r.download(), sc[0])
    r.upload(sc[1]); r.upload(sc[2]); assert np.array_equal(r.download(), sc[2])           # the unread upload is simply replaced
    r.upload(sc[0][:0]); assert r.n_points == 0 and len(r.download()) == 0
    # (2) finder, projector, factor
    f = api.CorrespondenceFinderProjective2f(ctx, _projector(361), 0.5, 0.8)
    r.upload(sc[1]); f.setFixed(r); f.setMoving(m); f.setLocalMapInSensor(wl.x0[1]); a = f.compute()
    f.setFixed(api.CloudSet(ctx, sc[1])); b = f.compute()
    assert len(a) > 50 and np.array_equal(a, b)
    r.upload(sc[2]); src, depth, _ = _projector(361).compute(ctx, r, np.zeros(3, np.float32))
    src2, depth2, _ = _projector(361).compute(ctx, sc[2], np.zeros(3, np.float32))
    assert np.array_equal(src, src2) and np.array_equal(depth, depth2)
    # (3) one alignment, one slice: k_align unpacks in its prologue and leaves the set behind
    al = _aligner(ctx, 361)
    for path in (0, 1, 2):                                     # automatic, one workgroup, split (the split path gets a launch of its own)
        r.upload(sc[0])
        ctx.set_option("align_path", path)
        try:
            g = al.compute_batch([r], [m], wl.x0[:1], want_stats=True)
        finally:
            ctx.set_option("align_path", 0)
        h = al.compute_batch([api.CloudSet(ctx, sc[0])], [m], wl.x0[:1], want_stats=True)
        assert g.status[0] == 0 and np.array_equal(g.pose, h.pose) and np.array_equal(g.information, h.information) and np.array_equal(g.stats, h.stats), path
        assert r.n_points == len(sc[0]) and np.array_equal(r.download(), sc[0]), path
    # (4) one alignment, two slices (k_align_pair), each with its own freshly uploaded scan; then the same set in both slices
    r2 = api.CloudSet.reserved(ctx, 2048)
    al2 = api.MultiAligner2D(ctx, max_iterations=8, min_num_inliers=10)
    for nc in (0.8, 0.7):
        al2.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector(361), 0.5, nc), min_num_correspondences=5))
    for same in (False, True):
        r.upload(sc[0]); r2.upload(sc[0] if same else sc[0][::2].copy())
        fx = [r, r] if same else [r, r2]
        g = al2.compute_batch(fx, [m, m], wl.x0[:1], want_stats=True)
        assert ctx.get_option("last_align_path") == 3
        ref = [api.CloudSet(ctx, sc[0]), api.CloudSet(ctx, sc[0] if same else sc[0][::2].copy())]
        h = al2.compute_batch(ref, [m, m], wl.x0[:1], want_stats=True)
        assert g.status[0] == 0 and np.array_equal(g.pose, h.pose) and np.array_equal(g.information, h.information) and np.array_equal(g.stats, h.stats), same
        assert np.array_equal(r.download(), sc[0]) and (same or np.array_equal(r2.download(), sc[0][::2]))
    # (5) more than one alignment, or another finder: the set is unpacked by a launch in front
    r.upload(sc[1])
    g = al.compute_batch([r], [m], np.stack([wl.x0[1], wl.x0[1]]))
    h = al.compute_batch([api.CloudSet(ctx, sc[1])], [m], np.stack([wl.x0[1], wl.x0[1]]))
    assert np.array_equal(g.pose, h.pose) and np.all(g.status == 0)
    aln = api.MultiAligner2D(ctx, max_iterations=5, min_num_inliers=10)
    aln.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.8), min_num_correspondences=5))
    r.upload(sc[2])
    g = aln.compute_batch([r], [m], wl.x0[2:3]); h = aln.compute_batch([api.CloudSet(ctx, sc[2])], [m], wl.x0[2:3])
    assert np.array_equal(g.pose, h.pose) and g.status[0] == 0
    # (6) the merger reads an uploaded measurement; the clipper replaces an uploaded output set
    scene = api.CloudSet.reserved(ctx, 20000); scene.upload(wl.map_points[:3000])
    pose = synth.invert_poses(wl.x_true[:1])[0].astype(np.float32)
    mg = api.MergerProjective2D(ctx, _projector(361), merge_threshold=0.2)
    r.upload(sc[0]); mg.setScene(scene); mg.setMeasurement(r); mg.setMeasurementInScene(pose); mg.compute()
    omap, _ = po.merge_scene(po.Projector(361, -math.pi, math.pi, 0.3, 30.0, 0.0), wl.map_points[:3000], sc[0], pose, 0.2)
    assert np.array_equal(scene.download(), omap)


def test_device_tensors_computed_a_moment_ago_are_read_complete(ctx, po):
    """Device-resident inputs are read on the context's own (non-blocking) stream: the Python mirror waits for the stream that produced the
    tensor (include/lsm2d.h, ORDERING).  Ranges and map points that are the result of a long chain of GPU operations queued immediately
    before the call must come through complete."""
    import torch
    world = synth.make_world(2)
    poses = synth.sample_poses(world, 64, seed=21)
    a0, a1 = -2.34747, 2.35619
    ranges = synth.make_scan_ranges(world, poses, n_beams=1081, angle_min=a0, angle_max=a1, noise_sigma=0.0, seed=3)
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    pre.setRawData(ranges, a0, a1, 0.0, 30.0)
    want = [c for c in (pre.compute().download(i) for i in range(len(poses)))]
    base = torch.from_numpy(ranges).to("cuda:0")
    big = torch.randn(4096, 4096, device="cuda:0")
    for _ in range(3):
        junk = big
        for _ in range(12):
            junk = junk @ big * 1e-2                       # tens of milliseconds of queued work on torch's stream
        r = (base * 2.0 + junk[0, 0] * 0.0) * 0.5          # exact in fp32: == base, but only once the chain above has run
        pre.setRawData(r, a0, a1, 0.0, 30.0)
        cs = pre.compute()
        for i in (0, 31, 63):
            assert np.array_equal(cs.download(i), want[i])
    wl = synth.make_workload(2, 20000, seed=5)
    m = torch.from_numpy(wl.map_points).to("cuda:0")
    junk = big
    for _ in range(12):
        junk = junk @ big * 1e-2
    m2 = (m * 2.0 + junk[0, 0] * 0.0) * 0.5
    assert np.array_equal(api.CloudSet(ctx, m2).download(0), wl.map_points)


def test_culling_and_placement_change_no_bit(ctx, po):
    """The exact culling of the moving cloud against the fixed canvas (chunk_may_matter), both forms of the culled stream (units /
    row-major), and the balanced placement of a culled batch (k_cull_estimate / balance_order) change WHERE and WHETHER a map point is
    visited, never a result: poses, information matrices, iteration counts and every iteration's statistics are bit-identical to the
    plain stream -- on a batch that fills the chip, on a shuffled map (chunks without locality: nothing is culled), on a partial-FOV
    canvas, with the Cauchy kernel, and with a second slice."""
    wl = synth.make_workload(600, 60000, seed=8)
    shuffled = wl.map_points[np.argsort(synth.Stream(3).uniform(len(wl.map_points)))]
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    def run(al, moving_sets, **opts):      # None: a variant only the experiments build of the library has
        try:
            if not xset(ctx, **opts):
                return None
            return al.compute_batch([fixed] * len(moving_sets), moving_sets, wl.x0, want_stats=True)
        finally:
            xset(ctx, cull=1, balance=1, xcd_lockstep=0, cull_block=0, proj_modes=1, balance_notes=1, two_stage=0, cull_est_um=0, cull_est_urad=40000, estimate_reuse=1, cull_keep=1)
    for name, mp in (("ordered", wl.map_points), ("shuffled", shuffled)):
        moving = api.CloudSet(ctx, mp)
        for tag, al in (("plain", _aligner(ctx)), ("cauchy 270 deg", _aligner(ctx, robustifier=api.RobustifierCauchy(0.05)))):
            if tag != "plain":      # a partial field of view: columns outside the canvas never hold a fixed point
                al.param_slice_processors[0].param_finder.param_projector = api.PointNormal2fProjectorPolar(811, -0.75 * math.pi, 0.75 * math.pi, 0.3, 25.0)
            ref = run(al, [moving], cull=0)
            # (proj_modes 0: the shared instantiation instead of the one with the culled stream only)
            # (round 4: the placement groups workgroup ids by the CU the previous launch of the same shape ran them on -- the second and third plain
            # runs below place by the first one's notes --, "balance_notes" 0: by the round-3 assumption; other margins in the work estimate)
            # (round 5: the third plain run finds the second one's batch unchanged and keeps its placement -- no estimate launch; "estimate_reuse" 0: made afresh;
            # everything from "estimate_reuse" on lives in the experiments build only)
            for opts in (dict(cull=1), dict(cull=1), dict(cull=1), dict(cull=1, balance=0), dict(cull=1, estimate_reuse=0), dict(cull=1, xcd_lockstep=1),
                         dict(cull=1, two_stage=1), dict(cull=1, cull_est_um=60000, cull_est_urad=0), dict(cull=1, balance_notes=0),
                         dict(cull=1, two_stage=1, balance_notes=0), dict(cull=2), dict(cull=1, cull_block=6), dict(cull=1, cull_block=98), dict(cull=1, proj_modes=0), dict(cull=1, cull_keep=0)):
                got = run(al, [moving], **opts)
                if got is None:
                    continue
                assert np.array_equal(got.pose, ref.pose) and np.array_equal(got.information, ref.information), (name, tag, opts)
                assert np.array_equal(got.status, ref.status) and np.array_equal(got.iterations, ref.iterations) and np.array_equal(got.stats, ref.stats), (name, tag, opts)
    # two projective slices (the same clouds twice, different gates): culling per slice
    al2 = _aligner(ctx, its=10)
    al2.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector(721), 0.3, 0.9), min_num_correspondences=10))
    moving = api.CloudSet(ctx, wl.map_points)
    ref = run(al2, [moving, moving], cull=0)
    got = run(al2, [moving, moving], cull=1)
    assert np.array_equal(got.pose, ref.pose) and np.array_equal(got.information, ref.information) and np.array_equal(got.stats, ref.stats)
    # ... and the oracle agrees bit for bit with the culled run
    for i in (0, 300, 599):
        sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        rt = po.align(po.aligner_params(10, device_order=True), [po.slice_params(), po.slice_params(canvas_cols=721, point_distance=0.3, normal_cos=0.9)],
                      [sc, sc], [wl.map_points, wl.map_points], wl.x0[i])
        _assert_bitwise_equal_to_device_order_oracle(got, i, rt, ("two slices culled", i))
    # point-query finders in the tracker's wiring (a tree / grid per scan, every map point a query): tiles of 64 map points with no scan point
    # within reach are skipped -- the same bits with and without, ordered and shuffled map, a map with non-finite points, a far-off start pose
    broken = wl.map_points.copy(); broken[5000, 0] = np.nan; broken[20000:20003, 1] = np.inf
    far = wl.x0.copy(); far[::7, 0] += 300.0
    for finder in (api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.8, search="exact"),
                   api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.25, normal_cos=0.8, search="kdtree"),
                   api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.002, normal_cos=0.8, search="exact")):
        al = api.MultiAligner2D(ctx, max_iterations=6, min_num_inliers=10)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10, robustifier=api.RobustifierCauchy(0.05)))
        for name, mp, x0 in (("ordered", wl.map_points, wl.x0), ("shuffled", shuffled, wl.x0), ("non-finite", broken, wl.x0), ("far", wl.map_points, far)):
            moving = api.CloudSet(ctx, mp)
            res = {}
            for cull in (0, 1):
                ctx.set_option("cull", cull)
                try:
                    res[cull] = al.compute_batch([fixed], [moving], x0, want_stats=True)
                    assert ctx.get_option("last_query_cull") == cull
                finally:
                    ctx.set_option("cull", 1)
            a, c = res[0], res[1]
            assert np.array_equal(a.pose, c.pose, equal_nan=True) and np.array_equal(a.information, c.information, equal_nan=True), (finder.search, name)
            assert np.array_equal(a.status, c.status) and np.array_equal(a.iterations, c.iterations) and np.array_equal(a.stats, c.stats), (finder.search, name)
            if name == "ordered" and finder.param_max_distance_m > 0.1:
                assert (a.status == 0).mean() > 0.9, (finder.search, (a.status == 0).mean())
            if name in ("ordered", "far"):      # the instantiations with one form of the search only (grid NN without the search in global memory, KD-tree with the whole tree in LDS) against the shared ones
                if xset(ctx, nn_lds_only=0, kd_modes=0):
                    try:
                        s0 = al.compute_batch([fixed], [moving], x0, want_stats=True)
                    finally:
                        xset(ctx, nn_lds_only=1, kd_modes=1)
                    assert np.array_equal(s0.pose, c.pose, equal_nan=True) and np.array_equal(s0.information, c.information, equal_nan=True) and np.array_equal(s0.stats, c.stats), name


def test_stream_pipeline_begin_wait_and_refill_equal_the_synchronous_calls(ctx, po):
    """Round 5: the streaming form of the path -- fresh LaserMessage batches every step (raw_data_preprocessor_projective_2d.cpp:13-51 feeding the aligner of
    apps/visual_test_aligner_2d.cpp:123-156) -- lsm2d_preprocess_scans_refill into one of two alternating scan sets, lsm2d_align_batch_begin for step i while
    step i - 1 is still in flight (its pre-kernels on the context's second stream), lsm2d_align_batch_wait one step behind.  Every step's poses, information
    matrices, statuses, iteration counts and statistics are BITWISE those of the synchronous calls on the same ranges (lsm2d_preprocess_scans +
    lsm2d_align_batch), the refilled clouds are the oracle's, and the oracle's aligner (device order) reproduces sampled alignments bit for bit.  Also: the
    third batch in flight is refused, a batch of another size can follow, and the context is clean afterwards (a synchronous call still works)."""
    world = synth.make_world(5)
    a0, a1 = -2.34747, 2.35619
    n, beams, n_batches = 300, 721, 3
    m = synth.make_map(world, 60000)
    mset = api.CloudSet(ctx, m)
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    pp = po.Preprocessor(beams, a0, a1, 0.3, 20.0, 0.3, 5, 0.02)
    al = api.MultiAligner2D(ctx, max_iterations=12, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(
        api.CorrespondenceFinderProjective2f(ctx, api.PointNormal2fProjectorPolar(beams, -math.pi, math.pi, 0.3, 20.0)), min_num_correspondences=10))
    batches = []
    for k in range(n_batches):
        poses = synth.sample_poses(world, n, seed=40 + k)
        rg = synth.make_scan_ranges(world, poses, n_beams=beams, angle_min=a0, angle_max=a1, noise_sigma=0.003, seed=k)
        if k == 1:
            rg[7, :] = 0.01; rg[11, 200:400] = np.inf      # an empty cloud and a gap: ragged sizes that stay on the device
        x_true, x0 = synth.initial_guesses(poses, seed=50 + k)
        pre.setRawData(rg, a0, a1, 0.0, 30.0)
        fixed = pre.compute()
        want = al.compute_batch([fixed], [mset], x0.astype(np.float32), want_stats=True)
        batches.append((rg, x0.astype(np.float32), x_true, want, fixed.counts.copy()))
        fixed.close()
    # the pipeline: two scan sets, two prepared batches, one step in flight
    pre.setRawData(batches[0][0], a0, a1, 0.0, 30.0); set_a = pre.compute()
    pre.setRawData(batches[1][0], a0, a1, 0.0, 30.0); set_b = pre.compute()
    sets = (set_a, set_b)
    prep = (al.prepare_batch([set_a], [mset], batches[0][1], want_stats=True), al.prepare_batch([set_b], [mset], batches[1][1], want_stats=True))
    uploads0 = ctx.get_option("uploads")
    steps, got = 7, {}
    for i in range(steps):
        k = i % n_batches
        pre.setRawData(batches[k][0], a0, a1, 0.0, 30.0)
        pre.refill(sets[i % 2])
        prep[i % 2].set_init_poses(batches[k][1])
        prep[i % 2].begin()
        if i > 0 and (i - 1) not in got:
            got[i - 1] = prep[(i - 1) % 2].wait(copy=True)
        if i == 2:      # two batches in flight (step 2 and one more reading the OTHER set, which holds step 1's scans); a third is refused; waits in the order of the begins
            extra = al.prepare_batch([sets[1]], [mset], batches[1][1], want_stats=True)
            extra.begin()
            third = al.prepare_batch([sets[1]], [mset], batches[1][1])
            with pytest.raises(Exception):
                third.begin()
            with pytest.raises(Exception):      # ... and so is anything else that would stage data through the context: both lanes' buffers belong to the batches in flight
                pre.compute()
            got[2] = prep[0].wait(copy=True)
            ex = extra.wait(copy=True)
            assert np.array_equal(ex.pose, batches[1][3].pose) and np.array_equal(ex.stats, batches[1][3].stats)
    got[steps - 1] = prep[(steps - 1) % 2].wait(copy=True)
    assert ctx.get_option("uploads") - uploads0 == steps
    for i in range(steps):
        rg, x0, x_true, want, counts = batches[i % n_batches]
        g = got[i]
        assert np.array_equal(g.pose, want.pose) and np.array_equal(g.information, want.information), i
        assert np.array_equal(g.status, want.status) and np.array_equal(g.iterations, want.iterations) and np.array_equal(g.stats, want.stats), i
    # what the last refill left in its set: the oracle's clouds, sizes read from the device on demand
    last = sets[(steps - 1) % 2]; rg, x0, x_true, want, counts = batches[(steps - 1) % n_batches]
    for c in (0, 7, 11, n - 1):
        assert np.array_equal(last.download(c), po.preprocess_scan(pp, rg[c])), c
    # ... and the oracle's aligner on the oracle's clouds, device order: bit for bit
    for c in (0, 150, n - 1):
        sc = po.preprocess_scan(pp, rg[c])
        rt = po.align(po.aligner_params(12, device_order=True), [po.slice_params(canvas_cols=beams, range_max=20.0)], [sc], [m], x0[c])
        assert np.array_equal(want.pose[c], rt["pose"]) and np.array_equal(want.information[c], rt["H"]), c
    # (the clouds are PCA normals on 2 cm voxels of noisy ranges, one batch with gaps: centimetres for nearly all, not 1e-4 -- the bits above are the gate)
    ok = want.status == 0
    err = np.abs(want.pose - x_true)[ok]
    assert ok.mean() > 0.95 and np.percentile(err[:, :2].max(1), 95) < 3e-2 and np.percentile(err[:, 2], 95) < 1e-2
    # the context is as it was: a synchronous call, another size
    small = al.compute_batch([last], [mset], x0, want_stats=True)
    assert np.array_equal(small.pose, want.pose) and np.array_equal(small.stats, want.stats)
    # the order include/lsm2d.h recommends: THREE scan sets, per step  begin(i) ; refill(set of step i + 1) ; wait(i - 1)  -- while a batch is in flight the refill's
    # copy and its preprocessing run on streams of their own, joined by the begin() that reads the set.  Step 4 refills its set TWICE (other ranges first): the
    # second copy must wait for the first launch, which still reads the set's range buffer (the set's own event)
    pre.setRawData(batches[2][0], a0, a1, 0.0, 30.0); set_c = pre.compute()
    sets3 = (set_a, set_b, set_c)
    prep3 = [al.prepare_batch([s_], [mset], batches[0][1], want_stats=True) for s_ in sets3]
    steps3, got3 = 8, {}
    pre.setRawData(batches[0][0], a0, a1, 0.0, 30.0); pre.refill(sets3[0])
    for i in range(steps3):
        prep3[i % 3].set_init_poses(batches[i % n_batches][1])
        prep3[i % 3].begin()
        if i + 1 < steps3:
            if i + 1 == 4:
                pre.setRawData(batches[(i + 2) % n_batches][0], a0, a1, 0.0, 30.0); pre.refill(sets3[(i + 1) % 3])
            pre.setRawData(batches[(i + 1) % n_batches][0], a0, a1, 0.0, 30.0); pre.refill(sets3[(i + 1) % 3])
        if i > 0:
            got3[i - 1] = prep3[(i - 1) % 3].wait(copy=True)
    got3[steps3 - 1] = prep3[(steps3 - 1) % 3].wait(copy=True)
    for i in range(steps3):
        want_i = batches[i % n_batches][3]; g = got3[i]
        assert np.array_equal(g.pose, want_i.pose) and np.array_equal(g.information, want_i.information) and np.array_equal(g.status, want_i.status), i
        assert np.array_equal(g.iterations, want_i.iterations) and np.array_equal(g.stats, want_i.stats), i
    ctx.synchronize()
    for c in (0, 7, 11, n - 1):      # (lsm2d_synchronize covers the side streams; the last refill's clouds are the oracle's)
        assert np.array_equal(sets3[(steps3 - 1) % 3].download(c), po.preprocess_scan(pp, batches[(steps3 - 1) % n_batches][0][c])), c
    for s_ in sets3:
        s_.close()


def test_first_call_of_a_fresh_context_is_an_asynchronous_begin_and_two_batches_overlap(small_workload):
    """Round 5: an asynchronously begun batch launches on its LANE's own stream, so that the younger of two batches in flight fills the slots the older one's tail
    leaves free.  With nothing in flight a batch's start poses and estimate are queued on the context's own stream: the lane's stream must wait for them (the first
    build did not: on a fresh context -- nothing valid in the lane's scratch yet -- four alignments in five never reported).  A context of its own, first call a
    begin(); then two batches of 320 alignments (index arrays over six scans, other poses) alternating, two in flight: every result BITWISE the synchronous call's."""
    wl = small_workload
    c = api.Context(0)
    try:
        al = _aligner(c)
        fixed = api.CloudSet(c, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c, wl.map_points)
        n = 320
        fi = (np.arange(n, dtype=np.int32) % len(wl.x0)).reshape(1, n)
        rng = np.random.default_rng(17)
        xa = (wl.x0[fi[0]] + rng.normal(0, [0.02, 0.02, 0.005], (n, 3))).astype(np.float32)
        xb = (wl.x0[fi[0]] + rng.normal(0, [0.02, 0.02, 0.005], (n, 3))).astype(np.float32)
        pa = al.prepare_batch([fixed], [moving], xa, fixed_index=fi, want_stats=True)
        pb = al.prepare_batch([fixed], [moving], xb, fixed_index=fi, want_stats=True)
        pa.begin()                                  # the context's very first aligner call
        pb.begin()                                  # ... and a second one behind it: two lanes, two streams
        ra = pa.wait(copy=True); rb = pb.wait(copy=True)
        wa = al.compute_batch([fixed], [moving], xa, fixed_index=fi, want_stats=True)
        wb = al.compute_batch([fixed], [moving], xb, fixed_index=fi, want_stats=True)
        for g, w in ((ra, wa), (rb, wb)):
            assert np.array_equal(g.pose, w.pose) and np.array_equal(g.information, w.information) and np.array_equal(g.status, w.status)
            assert np.array_equal(g.iterations, w.iterations) and np.array_equal(g.stats, w.stats)
        assert (wa.status == 0).all() and not np.array_equal(wa.pose, wb.pose)
        got = list(api.run_pipelined([pa, pb, pa, pb, pa]))      # the same as a generator over a queue of batches
        assert len(got) == 5
        for k, g in enumerate(got):
            w = (wa, wb)[k & 1]
            assert np.array_equal(g.pose, w.pose) and np.array_equal(g.stats, w.stats) and np.array_equal(g.status, w.status), k
        for lane_streams in (1, 0):                 # (0: every launch in order on the context's stream, as first built -- a knob of the experiments build)
            if not xset(c, lane_streams=lane_streams):
                continue
            for k in range(6):                      # a pipeline of them: begin(k) ; wait(k - 1)
                (pa, pb)[k & 1].begin()
                if k:
                    g = (pa, pb)[(k - 1) & 1].wait(copy=True); w = (wa, wb)[(k - 1) & 1]
                    assert np.array_equal(g.pose, w.pose) and np.array_equal(g.stats, w.stats), (lane_streams, k)
            g = pb.wait(copy=True)
            assert np.array_equal(g.pose, wb.pose) and np.array_equal(g.stats, wb.stats)
        xset(c, lane_streams=1)
        # a SYNCHRONOUS call while a begun batch is on the chip (the other lane; its estimate shares the ticket counter with the begun batch's: ordered behind it)
        xc = (wl.x0[fi[0]] + rng.normal(0, [0.02, 0.02, 0.005], (n, 3))).astype(np.float32)
        wc = al.compute_batch([fixed], [moving], xc, fixed_index=fi, want_stats=True)
        pa.set_init_poses(xc + np.float32(0.001)); pa.begin()
        gc_ = al.compute_batch([fixed], [moving], xc, fixed_index=fi, want_stats=True)
        pa.wait()
        assert np.array_equal(gc_.pose, wc.pose) and np.array_equal(gc_.stats, wc.stats) and np.array_equal(gc_.status, wc.status)
        fixed.close(); moving.close()
    finally:
        c.close()


def test_asynchronous_entry_points_reject_what_they_must_and_survive_abandonment(small_workload):
    """Edge cases of lsm2d_align_batch_begin / _wait / lsm2d_preprocess_scans_refill through the raw ABI: null arguments, an empty batch (begun and waited for: a
    no-op), wait's outputs missing (the batch is still retired: the lane is free again), a refill into a set of another shape or another context, and a context
    destroyed while a begun batch was never waited for (its streams are drained, nothing is touched afterwards)."""
    import ctypes as C
    from srrg2_laser_slam_2d_amd import _capi
    wl = small_workload
    c = api.Context(0)
    lib = c._lib
    al = _aligner(c)
    fixed = api.CloudSet(c, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c, wl.map_points)
    pb = al.prepare_batch([fixed], [moving], wl.x0)
    h = C.c_void_p()
    assert lib.lsm2d_align_batch_begin(None, C.byref(pb._ap), C.byref(pb._b), 0, C.byref(h)) == _capi.BAD_ARGUMENT
    assert lib.lsm2d_align_batch_begin(c.handle, C.byref(pb._ap), C.byref(pb._b), 0, None) == _capi.BAD_ARGUMENT
    assert lib.lsm2d_align_batch_wait(None, None, None, None, None, None) == _capi.BAD_ARGUMENT
    # an empty batch: begun, waited for, nothing happens
    pe = al.prepare_batch([fixed], [moving], np.zeros((0, 3), np.float32))
    pe.begin(); re = pe.wait()
    assert len(re.pose) == 0
    # wait without outputs: an error, but the batch is retired -- two more can be begun and give the right answer
    want = al.compute_batch([fixed], [moving], wl.x0)
    check_rc = lib.lsm2d_align_batch_begin(c.handle, C.byref(pb._ap), C.byref(pb._b), 0, C.byref(h))
    assert check_rc == 0 and h.value
    assert lib.lsm2d_align_batch_wait(h, None, None, None, None, None) == _capi.BAD_ARGUMENT
    pb2 = al.prepare_batch([fixed], [moving], wl.x0)
    pb.begin(); pb2.begin()
    assert np.array_equal(pb.wait().pose, want.pose) and np.array_equal(pb2.wait().pose, want.pose)
    # refill: the set must come from lsm2d_preprocess_scans with the same number of scans and beams, on this context
    world = synth.make_world(1); poses = synth.sample_poses(world, 9, seed=2)
    a0, a1 = -2.0, 2.0
    rg = synth.make_scan_ranges(world, poses, n_beams=361, angle_min=a0, angle_max=a1, seed=3)
    pre = api.RawDataPreprocessorProjective2D(c, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    pre.setRawData(rg, a0, a1, 0.0, 30.0); sset = pre.compute()
    pre.setRawData(rg[:5], a0, a1, 0.0, 30.0)
    with pytest.raises(Exception):
        pre.refill(sset)                                   # 5 scans into a set of 9
    rg2 = synth.make_scan_ranges(world, poses, n_beams=181, angle_min=a0, angle_max=a1, seed=3)
    pre.setRawData(rg2, a0, a1, 0.0, 30.0)
    with pytest.raises(Exception):
        pre.refill(sset)                                   # other beams
    with pytest.raises(Exception):
        pre.setRawData(rg, a0, a1, 0.0, 30.0); pre.refill(fixed)      # a set that no preprocessor made
    c2 = api.Context(0)
    pre2 = api.RawDataPreprocessorProjective2D(c2, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    pre2.setRawData(rg, a0, a1, 0.0, 30.0)
    with pytest.raises(Exception):
        pre2.refill(sset)                                  # another context's set
    pre.setRawData(rg, a0, a1, 0.0, 30.0); pre.refill(sset)      # ... and the right one still works
    assert int(sset.counts.sum()) > 0
    # a context destroyed with a begun batch that nobody waits for
    al2 = _aligner(c2)
    f2 = api.CloudSet(c2, wl.scan_points, wl.scan_offsets); m2 = api.CloudSet(c2, wl.map_points)
    lost = al2.prepare_batch([f2], [m2], wl.x0)
    lost.begin()
    c2.close()
    # the first context is untouched by all of it
    got = al.compute_batch([fixed], [moving], wl.x0)
    assert np.array_equal(got.pose, want.pose)
    c.close()


def test_prepared_batch_equals_compute_batch(ctx, small_workload):
    """MultiAligner2D.prepare_batch: the descriptor and the result arrays built once, lsm2d_align_batch called again and again (what bench.py times) --
    the same results as compute_batch, call after call, also after new start poses were written in place."""
    wl = small_workload
    al = _aligner(ctx)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    want = al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
    prep = al.prepare_batch([fixed], [moving], wl.x0, want_stats=True)
    for _ in range(3):
        got = prep.run()
        assert np.array_equal(got.pose, want.pose) and np.array_equal(got.information, want.information) and np.array_equal(got.status, want.status)
        assert np.array_equal(got.iterations, want.iterations) and np.array_equal(got.stats, want.stats)
    x1 = wl.x0.copy(); x1[:, 0] += 0.01
    prep.set_init_poses(x1)
    got = prep.run(); want1 = al.compute_batch([fixed], [moving], x1, want_stats=True)
    assert np.array_equal(got.pose, want1.pose) and np.array_equal(got.stats, want1.stats) and not np.array_equal(want1.pose, want.pose)
    # round 5: a batch that comes again with the same input block is not uploaded again -- unless something else used the context's scratch in between
    # (a finder call, another batch), or one start pose differs by one bit
    n = 300
    fi = (np.arange(n, dtype=np.int32) % len(wl.x0)).reshape(1, n)
    xa = wl.x0[fi[0]].astype(np.float32).copy()
    pa = al.prepare_batch([fixed], [moving], xa, fixed_index=fi, want_stats=True)
    wa = al.compute_batch([fixed], [moving], xa, fixed_index=fi, want_stats=True)
    f0 = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    finder = api.CorrespondenceFinderProjective2f(ctx, _projector())
    for k in range(6):
        if k == 2:                                   # other users of the scratch in between
            finder.setFixed(f0); finder.setMoving(wl.map_points); finder.setLocalMapInSensor(wl.x0[0]); finder.compute()
        if k == 4:
            al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
        g = pa.run()
        assert np.array_equal(g.pose, wa.pose) and np.array_equal(g.stats, wa.stats) and np.array_equal(g.status, wa.status), k
    xb = xa.copy(); xb[7, 2] = np.nextafter(xb[7, 2], np.float32(10.0))
    pa.set_init_poses(xb)
    wb = al.compute_batch([fixed], [moving], xb, fixed_index=fi, want_stats=True)
    g = pa.run()
    assert np.array_equal(g.pose, wb.pose) and np.array_equal(g.stats, wb.stats)
    pa.set_init_poses(xa)
    g = pa.run()
    assert np.array_equal(g.pose, wa.pose) and np.array_equal(g.stats, wa.stats)


def test_two_launches_for_one_batch_change_no_bit(ctx, po):
    """Round 4 (late): a culled batch of about one dispatch round CAN run as two launches ("two_stage" 1; measured, slower, off by default: DESIGN
    App. A) -- iteration 0 of every alignment anywhere on the chip (k_first_iteration), then the remaining iterations placed by the length of
    iteration 1's unit lists -- with pose, information matrix, phase and termination state carried in memory between them.  Against the single
    launch, bit for bit: poses, information matrices, statuses,
    iteration counts, every iteration's statistics and digest -- with the termination criterion, the inlier-only runs, the Cauchy kernel, two slices, the
    shortest loop that is split at all (4 iterations), start poses that fail in iteration 0 (they finish in the first launch) -- and the oracle agrees."""
    need_experiments(ctx)
    wl = synth.make_workload(300, 60000, seed=21)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    x0 = wl.x0.copy(); x0[::11, 0] += 250.0; x0[5::17, 2] += 1.2      # some alignments start beyond the map / badly rotated
    cases = []
    al = _aligner(ctx, its=20); cases.append(("plain 20", al, 1, dict()))
    al = _aligner(ctx, its=4); cases.append(("4 iterations", al, 1, dict()))
    al = _aligner(ctx, its=12, robustifier=api.RobustifierCauchy(0.02)); al.param_termination_chi_epsilon = 1e-3
    al.param_enable_inlier_only_runs = True; al.param_keep_only_inlier_correspondences = True
    cases.append(("Cauchy + epsilon + inlier runs", al, 1, dict(robustifier=po.ROBUST_CAUCHY, chi_threshold=0.02)))
    al = _aligner(ctx, its=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector(721), 0.3, 0.9), min_num_correspondences=10))
    cases.append(("two slices", al, 2, dict()))
    for name, al, ns, okw in cases:
        got = {}
        for ts in (1, 0):
            ctx.set_option("two_stage", ts)
            try:
                got[ts] = al.compute_batch([fixed] * ns, [moving] * ns, x0, want_stats=True)
                if ts == 1:
                    assert ctx.get_option("last_align_path") == 1
            finally:
                ctx.set_option("two_stage", 0)
        a, b = got[1], got[0]
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.information, b.information), name
        assert np.array_equal(a.status, b.status) and np.array_equal(a.iterations, b.iterations) and np.array_equal(a.stats, b.stats), name
        assert (a.status != 0).any() and (a.status == 0).mean() > 0.6, (name, (a.status == 0).mean())
        if name in ("plain 20", "Cauchy + epsilon + inlier runs"):
            ap_ = po.aligner_params(al.param_max_iterations, device_order=True, termination_chi_epsilon=al.param_termination_chi_epsilon,
                                    enable_inlier_only_runs=al.param_enable_inlier_only_runs, keep_only_inlier_correspondences=al.param_keep_only_inlier_correspondences)
            for i in (0, 5, 11, 150, 299):
                sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
                rt = po.align(ap_, [po.slice_params(**okw)], [sc], [wl.map_points], x0[i])
                _assert_bitwise_equal_to_device_order_oracle(a, i, rt, (name, i))


@pytest.mark.gpu
def test_latency_kernel_every_cloud_placement_equals_the_fused_kernel(ctx, po):
    """k_align_pair keeps a moving cloud of <= 1024 points and a fixed cloud of <= 4096 points per slice in LDS rows and has a walk of its
    own when both are there; each of the four combinations (and the sizes around the limits: 512 / 513 / 1024 / 1025 moving points, one
    and two slices, Cauchy, prior, a canvas with three columns per thread) must give k_align's bits -- poses, information matrices,
    statuses, iteration counts and per-iteration statistics -- and the oracle's in the device's order."""
    world = synth.make_world(11)
    robot = synth.sample_poses(world, 1, seed=31)
    big = synth.make_map(world, 20000, noise_sigma=0.003, seed=4)

    def in_robot_frame(cloud, pose):          # world cloud -> the frame of `pose` (fp64 arithmetic, rounded once: just another input)
        T = np.linalg.inv(synth.v2t(pose)); R = T[:2, :2]
        out = np.empty_like(cloud)
        out[:, :2] = (cloud[:, :2].astype(np.float64) @ R.T + T[:2, 2]).astype(np.float32)
        out[:, 2:] = (cloud[:, 2:].astype(np.float64) @ R.T).astype(np.float32)
        return np.ascontiguousarray(out)

    def scan(n_beams, seed, dpose=(0.0, 0.0, 0.0)):
        p = synth.compose_poses(robot, np.array([dpose]))
        pts, _ = synth.make_scans(world, p, n_beams=n_beams, noise_sigma=0.004, seed=seed)
        return pts

    big_local = in_robot_frame(big, robot[0])
    cases = []
    for n_mov in (300, 512, 513, 1024, 1025):                                      # moving on chip up to 1024, one or two points per thread
        mv = scan(1400, 7)[:n_mov]
        assert len(mv) == n_mov
        cases.append(("moving %d / fixed scan" % n_mov, [scan(900, 3)], [mv], 1))
    cases.append(("moving scan / fixed 20000 (no room in LDS)", [big_local], [scan(700, 5)], 1))
    cases.append(("moving 20000 / fixed scan", [scan(1000, 9)], [big_local], 1))
    cases.append(("moving 20000 / fixed 20000", [big_local], [big_local[::-1].copy()], 1))
    cases.append(("two slices: on chip + moving in memory", [scan(800, 13), scan(600, 14)], [scan(700, 15), big_local], 2))
    cases.append(("two slices, both on chip", [scan(721, 16), scan(500, 17)], [scan(640, 18), scan(900, 19)], 2))
    checked = 0
    for name, fixed, moving, ns in cases:
        for cols, use_prior in ((721, True), (1300, False)):
            al = api.MultiAligner2D(ctx, max_iterations=7, min_num_inliers=5)
            oslices = []
            for s in range(ns):
                proj = api.PointNormal2fProjectorPolar(cols + 60 * s, -math.pi, math.pi, 0.3, 25.0)
                f = api.CorrespondenceFinderProjective2f(ctx, proj, 0.6, 0.7)
                rob = api.RobustifierCauchy(0.02) if s == 0 else None
                S = np.float32([0.1, -0.05, 0.2]) if s == 1 else np.zeros(3, np.float32)
                sl = (api.AlignerSliceProcessorLaser2DWithSensor(f, sensor_in_robot=S, robustifier=rob, min_num_correspondences=3) if S.any()
                      else api.AlignerSliceProcessorLaser2D(f, robustifier=rob, min_num_correspondences=3))
                al.param_slice_processors.append(sl); oslices.append(_oracle_slice(po, sl.slice_params()))
            x0 = np.float32([[0.03, -0.02, 0.01]])
            pri = [(x0[0].copy(), np.diag([40.0, 30.0, 20.0]).astype(np.float32))] if use_prior else None
            fs = [api.CloudSet(ctx, c) for c in fixed]; ms = [api.CloudSet(ctx, c) for c in moving]
            res = {}
            for path in (1, 3):
                ctx.set_option("align_path", path)
                try:
                    res[path] = al.compute_batch(fs, ms, x0, priors=pri, want_stats=True)
                    assert ctx.get_option("last_align_path") == path
                finally:
                    ctx.set_option("align_path", 0)
            a, c = res[1], res[3]
            assert np.array_equal(a.pose, c.pose) and np.array_equal(a.information, c.information) and np.array_equal(a.status, c.status) and \
                np.array_equal(a.iterations, c.iterations), (name, cols)
            assert np.array_equal(a.stats[0][: a.iterations[0]], c.stats[0][: c.iterations[0]]), (name, cols, "statistics")
            assert a.stats[0]["n_correspondences"][0] > 20, (name, cols, "the case must form pairs")
            kw = dict(prior_z=pri[0][0], prior_omega=pri[0][1]) if use_prior else {}
            rt = po.align(po.aligner_params(7, min_num_inliers=5, device_order=True, **kw), oslices, fixed, moving, x0[0])
            _assert_bitwise_equal_to_device_order_oracle(c, 0, rt, (name, cols))
            checked += 1
    print("latency kernel: %d placements x canvases equal to k_align and to the device-order oracle bit for bit" % checked)


def test_latency_kernel_two_slices_one_empty_fixed_cloud_and_one_beyond_the_lds_rows(ctx, po):
    """Round-3 advisor finding: k_align_pair decided "fixed cloud on chip" per slice half, and the other side of that branch holds a barrier -- with a
    fixed cloud above 4 096 points in one slice (no LDS rows at all: pair_fix_cap == 0) and an EMPTY one in the other, only half of the workgroup
    executed it.  The predicate is workgroup-uniform now; the case runs, equals the fused kernel and the device-order mirror bit for bit."""
    world = synth.make_world(6)
    m = synth.make_map(world, 6000, seed=1)
    robots = synth.sample_poses(world, 1, seed=2)
    big, _ = synth.make_scans(world, robots, n_beams=5000, fov_deg=300.0)                 # a fixed cloud of ~5 000 points: beyond the 4 096 rows
    assert len(big) > 4096
    empty = np.zeros((0, 4), np.float32)
    x0 = synth.invert_poses(synth.compose_poses(robots, np.array([[0.03, -0.02, 0.02]]))).astype(np.float32)
    proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
    al = api.MultiAligner2D(ctx, max_iterations=6, min_num_inliers=5)
    for _ in range(2):
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), min_num_correspondences=3))
    osl = [_oracle_slice(po, s_.slice_params()) for s_ in al.param_slice_processors]
    for fixed in ([big, empty], [empty, big]):
        res = {}
        for path in (3, 1):
            ctx.set_option("align_path", path)
            try:
                res[path] = al.compute_batch([api.CloudSet(ctx, f) for f in fixed], [api.CloudSet(ctx, m)] * 2, x0, want_stats=True)
            finally:
                ctx.set_option("align_path", 0)
            assert ctx.get_option("last_align_path") == path
        a, c = res[1], res[3]
        assert np.array_equal(a.pose, c.pose) and np.array_equal(a.information, c.information) and np.array_equal(a.status, c.status) and np.array_equal(a.stats, c.stats)
        w = po.align(po.aligner_params(6, min_num_inliers=5, device_order=True), osl, fixed, [m, m], x0[0])
        _assert_bitwise_equal_to_device_order_oracle(c, 0, w, "one empty fixed cloud")
        assert c.status[0] == 0


def test_narrow_workgroups_keep_every_bit_and_the_width_rule(ctx, po):
    """Round 6: the culled projective stream in workgroups of 256 threads (k_align_narrow: six workgroups per CU instead of four, for batches just above a multiple
    of 1024 alignments).  The bin walk and the sums keep the wide kernel's 512 virtual threads: poses, information matrices, statistics and
    digests must equal the 512-thread kernel's bit for bit -- one and two slices, Cauchy, a prior, odd canvas sizes; and the automatic rule picks the width the
    batch-size sweep showed to pay."""
    wl = synth.make_workload(12, 60000, seed=31, map_noise=0.004, scan_noise=0.004)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    n = 300
    fi1 = (np.arange(n, dtype=np.int32) % 12).reshape(1, n)
    x0 = wl.x0[fi1[0]].astype(np.float32).copy(); x0[:, 1] += np.linspace(-0.02, 0.02, n, dtype=np.float32)
    pri = [(x0[i].copy(), np.diag([20.0, 30.0, 40.0]).astype(np.float32)) for i in range(n)]
    for ns, cols, rb in ((1, 1081, None), (1, 700, api.RobustifierCauchy(0.02)), (2, 721, api.RobustifierCauchy(0.02))):      # (two slices of 1081 columns do not fit the unit lists beside the canvases: the shared instantiation, one width)
        al = api.MultiAligner2D(ctx, max_iterations=12, min_num_inliers=10)
        for s in range(ns):
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector(cols - 180 * s)), min_num_correspondences=10, robustifier=rb))
        fi = np.tile(fi1, (ns, 1))
        res = {}
        for w in (512, 256):
            ctx.set_option("align_width", w); ctx.set_option("align_path", 1)
            try:
                res[w] = al.compute_batch([fixed] * ns, [moving] * ns, x0, fixed_index=fi, priors=pri if ns == 2 else None, want_stats=True)
                assert ctx.get_option("last_align_width") == w, (w, ctx.get_option("last_align_width"))
            finally:
                ctx.set_option("align_width", 0); ctx.set_option("align_path", 0)
        assert np.all(res[512].status == 0)
        for w in (256,):
            assert np.array_equal(res[w].pose, res[512].pose) and np.array_equal(res[w].information, res[512].information), (ns, cols, w)
            assert np.array_equal(res[w].status, res[512].status) and np.array_equal(res[w].iterations, res[512].iterations) and np.array_equal(res[w].stats, res[512].stats), (ns, cols, w)
        # ... and the 512-thread kernel is the device-order mirror's, as everywhere
        osl = [po.slice_params(canvas_cols=cols - 180 * s, robustifier=po.ROBUST_CAUCHY if rb else po.ROBUST_NONE, chi_threshold=0.02 if rb else 0.05) for s in range(ns)]
        for i in (0, 151, 299):
            c = int(fi1[0, i]); sc = wl.scan_points[wl.scan_offsets[c]:wl.scan_offsets[c + 1]]
            kw = dict(prior_z=pri[i][0], prior_omega=pri[i][1]) if ns == 2 else {}
            rt = po.align(po.aligner_params(12, device_order=True, **kw), osl, [sc] * ns, [wl.map_points] * ns, x0[i])
            _assert_bitwise_equal_to_device_order_oracle(res[256], i, rt, ("narrow", ns, cols, i))
    # the automatic rule (one slice, 1081 columns: 25 KB of LDS per workgroup)
    al = _aligner(ctx)
    for n_batch, want in ((1000, 512), (1024, 512), (1025, 1024), (1040, 1024), (1048, 1024), (1049, 256), (1100, 256), (1280, 256), (1400, 256), (1536, 256), (1537, 1024), (1700, 1024), (2047, 1024), (2048, 512), (2049, 1024), (2064, 1024), (2100, 512), (3071, 512), (3073, 1024), (3080, 1024), (3600, 512), (4095, 512)):
        fb = (np.arange(n_batch, dtype=np.int32) % 12).reshape(1, n_batch)
        ref = al.compute_batch([fixed], [moving], wl.x0[fb[0]], fixed_index=fb, want_stats=True)
        assert ctx.get_option("last_align_width") == want, (n_batch, ctx.get_option("last_align_width"), want)
        # (round 6, late) PACKED: the same batch in ONE dispatch round of 1024 workgroups, its lightest alignments two to a workgroup, one after the other (k_align_two; the
        # pairs are made by balance_order) -- automatic for 1025 .. 1048 and 1537 .. 2047 alignments, forced here for every size it can take; and against the wide kernel
        if 1024 < n_batch < 4096 and n_batch % 1024:
            got = {}
            for w in (512, 1024):
                try:
                    ctx.set_option("align_width", w)
                    got[w] = al.compute_batch([fixed], [moving], wl.x0[fb[0]], fixed_index=fb, want_stats=True)
                    assert ctx.get_option("last_align_width") == w, (n_batch, w, ctx.get_option("last_align_width"))
                finally:
                    ctx.set_option("align_width", 0)
            for w in (512, 1024):
                assert np.array_equal(got[w].pose, ref.pose) and np.array_equal(got[w].information, ref.information) and np.array_equal(got[w].status, ref.status), ("width", w, n_batch)
                assert np.array_equal(got[w].iterations, ref.iterations) and np.array_equal(got[w].stats, ref.stats), ("width", w, n_batch)
