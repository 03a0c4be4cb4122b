"""GPU parity tests, rows a2-a5: the polar projector and the four correspondence finders, pairs and order bit-exact against the oracle.

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


def test_projector_bit_exact(ctx, po, small_workload):
    wl = small_workload
    for cols, off in ((1081, 0.0), (721, 0.0), (360, 0.5)):
        pr = _projector(cols, off=off)
        for cloud, pose in ((wl.map_points, wl.x0[0]), (wl.map_points, wl.x_true[1].astype(np.float32)),
                            (wl.scan_points[wl.scan_offsets[2]:wl.scan_offsets[3]], np.zeros(3, np.float32)),
                            (wl.map_points[:1], wl.x0[0]), (wl.map_points[:7], wl.x0[0])):
            src, depth, xyn = pr.compute(ctx, cloud, pose)
            osrc, odepth, oxyn = po.project(po.Projector(cols, -math.pi, math.pi, 0.3, 30.0, off), cloud, pose)
            assert np.array_equal(src, osrc)
            assert np.array_equal(depth, odepth)
            assert np.array_equal(xyn[osrc >= 0], oxyn[osrc >= 0])


def test_projector_ties_lowest_index_and_empty(ctx, po):
    pr = _projector(360)
    pts = np.array([[2 * math.cos(0.5), 2 * math.sin(0.5), 1, 0]] * 5, np.float32)
    src, depth, _ = pr.compute(ctx, pts)
    assert (src >= 0).sum() == 1 and src[src >= 0][0] == 0
    # every point out of range -> empty canvas
    far = pts.copy(); far[:, :2] *= 100
    src, depth, _ = pr.compute(ctx, far)
    assert np.all(src == -1) and np.all(depth == np.finfo(np.float32).max)


@pytest.mark.parametrize("cols", [1081, 721])
def test_projective_finder_bit_exact(ctx, po, small_workload, cols):
    wl = small_workload
    finder = api.CorrespondenceFinderProjective2f(ctx, _projector(cols), point_distance=0.5, normal_cos=0.8)
    moving = api.CloudSet(ctx, wl.map_points)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    osp = po.slice_params(canvas_cols=cols)
    for i in range(len(wl.x0)):
        finder.setFixed(fixed, i); finder.setMoving(moving); finder.setLocalMapInSensor(wl.x0[i])
        got = finder.compute()
        want = po.find(osp, wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]], wl.map_points, wl.x0[i])
        assert len(want) > 100
        assert np.array_equal(got, want)      # same pairs, same (ascending column) order


def test_finder_reference_usage_errors(ctx):
    f = api.CorrespondenceFinderProjective2f(ctx, None)
    with pytest.raises(RuntimeError):
        f.compute()                              # Missing fixed! (correspondence_finder_projective_2d.cpp:25-27)
    f.setFixed(np.zeros((1, 4), np.float32)); f.setMoving(np.zeros((1, 4), np.float32))
    with pytest.raises(RuntimeError):
        f.compute()                              # Missing Projector (:21-23)


def test_short_divide_and_sqrt_sequences_are_correctly_rounded(tmp_path):
    """csrc/lsm2d_device.h forms the depth r = sqrt(r2) and the quotient min / r (the sine of the octant angle) from ONE v_rsq_f32 by
    short fused sequences; the oracle uses sqrtf and '/', so columns and depths are bit-exact only if those sequences round correctly
    for EVERY admissible input.  tools/fp_exact_check.hip proves it on the card: every fp32 in the range gate's [1e-30, 1e36] for the
    sqrt, and here a stride of 2^8 r2 mantissas (both exponent parities) x all 2^23 numerator mantissas for the quotient -- the full
    2^47 sweep (5 minutes of GPU) is profiles/r02/fp_exact_full_r02e.log: 4 inputs (r all ones, numerator a power of two) come out one
    ulp low, and the oracle's definition follows them.  The checker must also still catch the sequences known to be inexact."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "fp_exact_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-I" + os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                    "-o", exe, os.path.join(ROOT, "tools", "fp_exact_check.hip")], check=True, timeout=300)
    r = subprocess.run([exe, "32768"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = {ln.split()[0]: ln for ln in r.stdout.splitlines() if ln.startswith("  ")}
    assert "mismatches vs sqrtf: 0 " in lines["sqrt_rn_normal"]
    assert "bit mismatches: 0 " in lines["sincos_fixed"]              # the device rotates a pose with the host's (and the oracle's) bits
    rule = [ln for ln in r.stdout.splitlines() if "as the oracle defines it" in ln]
    assert rule and rule[0].rstrip().endswith(": 0 mismatches")          # the quotient sequence == the oracle's quotient, every input of the sample
    plain = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("div_by_depth ") and "mismatches vs n/r" in ln]
    assert plain and int(plain[0].split("mismatches vs n/r:")[1].split()[0]) <= 4      # ... and == IEEE n / r except the known ties
    assert "mismatches vs n/r: 0 " not in lines["D3(raw"]              # the check has teeth: the 3-operation quotient IS inexact


# ---- NN finder (CorrespondenceFinderKDTree2D, row a4) -------------------------------------------------------
@pytest.mark.parametrize("max_distance", [0.5, 0.05, 0.01])
def test_nn_finder_bit_exact_both_roles(ctx, po, small_workload, max_distance):
    wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    x = wl.x0[1] if max_distance >= 0.5 else wl.x_true[1].astype(np.float32)     # small gates need a near-true pose to match anything
    xb = synth.invert_poses(x[None, :].astype(np.float64))[0].astype(np.float32)
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=max_distance, normal_cos=0.8)
    for fixed, moving, pose in ((scan, wl.map_points, x), (wl.map_points, scan, xb)):     # role A (tracker wiring), role B (BASELINE wording)
        f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=max_distance, normal_cos=0.8)
        f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
        got = f.compute()
        want = po.find(osp, fixed, moving, pose)
        assert len(want) > 50
        assert np.array_equal(got, want)          # same pairs, ascending moving index


def test_nn_finder_edge_cases(ctx, po):
    rng = np.random.default_rng(5)
    def cloud(n, lo=-5, hi=5):
        p = rng.uniform(lo, hi, size=(n, 2)); a = rng.uniform(-np.pi, np.pi, n)
        return np.concatenate([p, np.cos(a)[:, None], np.sin(a)[:, None]], 1).astype(np.float32)
    fixed, moving = cloud(5000), cloud(3000, -7, 7)          # queries outside the fixed bounding box too
    fixed[10] = fixed[11]; moving[0, :2] = fixed[11, :2]     # duplicate fixed point: tie -> lowest index
    for md in (0.05, 0.3, 2.0):
        f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=md, normal_cos=-1.0)
        f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor([0.1, -0.2, 0.3])
        got = f.compute()
        want = po.find(po.slice_params(finder=po.FINDER_NN, max_distance=md, normal_cos=-1.0), fixed, moving, np.float32([0.1, -0.2, 0.3]), brute=True)
        assert np.array_equal(got, want)
    f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=-1.0)
    f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor([0, 0, 0])
    c = f.compute()
    assert c[0, 1] == 0 and c[0, 0] == 10
    # single fixed point, empty moving, all-identical fixed points
    one = fixed[:1]
    f.setFixed(one); f.setMoving(np.tile(one, (5, 1)))
    assert np.array_equal(f.compute(), np.stack([np.zeros(5, np.int32), np.arange(5, dtype=np.int32)], 1))
    f.setMoving(np.zeros((0, 4), np.float32))
    assert len(f.compute()) == 0
    f.setFixed(np.tile(one, (100, 1))); f.setMoving(one)
    assert np.array_equal(f.compute(), [[0, 0]])


# ---- distance-map finder (CorrespondenceFinderNN2D, row a5 / f4) ----------------------------------------------
def test_distmap_finder_bit_exact_and_aligner(ctx, po, small_workload):
    wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    for md, res in ((1.0, 0.05), (0.3, 0.1)):
        osp = po.slice_params(finder=po.FINDER_DISTMAP, max_distance=md, resolution=res)
        for fixed, moving, pose in ((scan, wl.map_points, wl.x0[1]),
                                    (wl.map_points, scan, synth.invert_poses(wl.x0[1:2].astype(np.float64))[0].astype(np.float32))):
            f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=md, resolution=res)
            f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
            got = f.compute()
            want = po.find(osp, fixed, moving, pose)
            assert len(want) > 100 and np.array_equal(got, want)
    # all-negative coordinates exercise the reference's bounding-box quirk (upper bound initialised to +FLT_MIN)
    neg = scan.copy(); neg[:, :2] -= np.float32([60, 60])
    f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.1)
    f.setFixed(neg); f.setMoving(neg[::3]); f.setLocalMapInSensor([0.01, 0.0, 0.0])
    assert np.array_equal(f.compute(), po.find(po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.1), neg, neg[::3], np.float32([0.01, 0, 0])))
    with pytest.raises(RuntimeError):
        api.CorrespondenceFinderNN2D(ctx, resolution=0.0).slice_params()
    # aligner with the distance-map finder (role A), vs oracle
    al = api.MultiAligner2D(ctx, max_iterations=20, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderNN2D(ctx, 1.0, 0.05), min_num_correspondences=10))
    al.setFixed({"points": scan}); al.setMoving({"points": wl.map_points}); al.setMovingInFixed(wl.x0[1])
    assert al.compute() == 0
    r = po.align(po.aligner_params(20), [po.slice_params(finder=po.FINDER_DISTMAP, max_distance=1.0, resolution=0.05)], [scan], [wl.map_points], wl.x0[1])
    d = np.abs(al.movingInFixed() - r["pose"])
    assert r["status"] == 0 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    assert al.iterationStats()["n_correspondences"][0] == r["stats"][0].n_corr


def test_nn_grid_build_one_workgroup_and_chip_wide_agree(ctx, po, small_workload):
    """The NN finder's grid over a map-sized cloud is built by chip-wide kernels (k_grid_big_*), over a scan-sized one by one workgroup
    (k_grid_build); option "grid_big_threshold" moves the border.  Both builds, on a scan, a 20k map and a 150k map (37 scan tiles), in a
    set that mixes sizes (and holds an empty cloud): the same pairs as the oracle, and the same aligner bits."""
    wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    big = synth.make_map(synth.make_world(5), 150000, noise_sigma=0.01, seed=3)
    inv = synth.invert_poses(wl.x0[1:2].astype(np.float64))[0].astype(np.float32)
    mixed_pts = np.concatenate([scan, wl.map_points, scan[:0], scan[::2]]).astype(np.float32)
    mixed_off = np.cumsum([0, len(scan), len(wl.map_points), 0, len(scan[::2])]).astype(np.int32)
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=0.4)
    want_map = po.find(osp, wl.map_points, scan, inv)
    want_scan = po.find(osp, scan, wl.map_points, wl.x0[1])
    want_big = po.find(osp, big, scan, inv)
    assert len(want_map) > 300 and len(want_scan) > 300
    results = []
    try:
        for thr in (1, 16384, 1 << 30):
            ctx.set_option("grid_big_threshold", thr)
            f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4)
            ms = api.CloudSet(ctx, mixed_pts, mixed_off)
            f.setMoving(scan); f.setLocalMapInSensor(inv); f.setFixed(ms, 1)
            assert np.array_equal(f.compute(), want_map), thr
            f.setFixed(ms, 2); assert len(f.compute()) == 0
            f.setMoving(wl.map_points); f.setLocalMapInSensor(wl.x0[1]); f.setFixed(ms, 0)
            assert np.array_equal(f.compute(), want_scan), thr
            f.setFixed(big); f.setMoving(scan); f.setLocalMapInSensor(inv)
            assert np.array_equal(f.compute(), want_big), thr
            al = api.MultiAligner2D(ctx, max_iterations=8, min_num_inliers=10)
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4), min_num_correspondences=10))
            r = al.compute_batch([api.CloudSet(ctx, wl.map_points)], [api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)],
                                 synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32))
            results.append((r.pose.copy(), r.information.copy()))
    finally:
        ctx.set_option("grid_big_threshold", 16384)
    for pose, info in results[1:]:
        assert np.array_equal(pose, results[0][0]) and np.array_equal(info, results[0][1])


def test_point_query_finder_one_workgroup_and_many_agree(ctx, po, small_workload):
    """lsm2d_find_correspondences with the NN and the distance-map finder runs on many workgroups when there are more queries than one
    workgroup takes in a trip (option "find_path" = 1 keeps it on one): same pairs, same (ascending moving index) order, and the oracle's --
    both roles, ragged sizes around the trip boundaries (256 queries with four lanes each, 1024 with one), no pair at all."""
    wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    inv = synth.invert_poses(wl.x0[1:2].astype(np.float64))[0].astype(np.float32)
    far = wl.map_points.copy(); far[:, :2] += np.float32([500, 500])
    cases = [(scan, wl.map_points, wl.x0[1]), (wl.map_points, scan, inv), (wl.map_points, scan[:257], inv), (wl.map_points, scan[:256], inv),
             (wl.map_points, scan[:513], inv), (scan, wl.map_points[:2049], wl.x0[1]), (scan, wl.map_points[:3072], wl.x0[1]), (scan, far, wl.x0[1])]
    try:
        for make, osp in ((lambda: api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4), po.slice_params(finder=po.FINDER_NN, max_distance=0.4)),
                          (lambda: api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05), po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.05))):
            for fixed, moving, pose in cases:
                want = po.find(osp, fixed, moving, pose)
                for mode in (0, 1):
                    ctx.set_option("find_path", mode)
                    f = make(); f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
                    assert np.array_equal(f.compute(), want), (osp.finder, len(fixed), len(moving), mode)
        # the projective finder z-buffers a map-sized cloud over many workgroups first (either side)
        big = synth.make_map(synth.make_world(3), 60000, noise_sigma=0.004, seed=5)
        osp = po.slice_params(canvas_cols=1081, range_max=30.0)
        for fixed, moving, pose in ((scan, np.concatenate([wl.map_points, big]), wl.x0[1]), (np.concatenate([wl.map_points, big]), scan, inv)):
            want = po.find(osp, fixed, moving, pose)
            for mode in (0, 1):
                ctx.set_option("find_path", mode)
                f = api.CorrespondenceFinderProjective2f(ctx, _projector()); f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
                assert len(want) > 50 and np.array_equal(f.compute(), want), (len(fixed), len(moving), mode)
    finally:
        ctx.set_option("find_path", 0)


def test_distmap_scatter_build_equals_gather_build_and_oracle(ctx, po, small_workload):
    """The distance maps are built from the points' side (k_distmap_stamp: one disc of atomic minima per point) unless the packed
    (d2, index) key does not fit; option "distmap_build" = 1 forces the per-pixel gather (k_distmap_fill).  Same pairs from both and from
    the oracle: many points per pixel (coarse pixels), a reach of zero pixels, a reach wider than the padding, a multi-cloud set, an
    empty cloud in the set, and a reach the scatter form cannot pack (falls back by itself)."""
    wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    empty_then_scans = np.concatenate([[0, 0], wl.scan_offsets[1:]]).astype(np.int32)       # cloud 0 empty, cloud 1 = scan 0 .. (offsets shifted by one cloud)
    cases = [(0.5, 0.05, scan, wl.map_points, wl.x0[1]), (0.4, 0.25, scan, wl.map_points, wl.x0[1]), (0.02, 0.05, scan, wl.map_points, wl.x0[1]),
             (4.5, 0.05, scan[::4], wl.map_points[::7], wl.x0[1]),                            # R = 90 pixels > half the padding (83): discs cross the border
             (0.5, 0.05, wl.map_points, scan, synth.invert_poses(wl.x0[1:2].astype(np.float64))[0].astype(np.float32)),
             (26.0, 0.1, wl.map_points, scan[::16], synth.invert_poses(wl.x0[1:2].astype(np.float64))[0].astype(np.float32))]   # R = 260 with 15 index bits: (d2, index) does not pack -> gather build under both settings
    try:
        for md, res, fixed, moving, pose in cases:
            want = po.find(po.slice_params(finder=po.FINDER_DISTMAP, max_distance=md, resolution=res), fixed, moving, pose)
            for mode in (0, 1):
                ctx.set_option("distmap_build", mode)
                f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=md, resolution=res)
                f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
                assert np.array_equal(f.compute(), want), (md, res, mode)
            assert len(want) > 0 or md < 0.05
        for mode in (0, 1):
            ctx.set_option("distmap_build", mode)
            fs = api.CloudSet(ctx, wl.scan_points, empty_then_scans)
            f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05)
            f.setMoving(wl.map_points); f.setLocalMapInSensor(wl.x0[1])
            f.setFixed(fs, 2)
            assert np.array_equal(f.compute(), po.find(po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.05), scan, wl.map_points, wl.x0[1]))
            f.setFixed(fs, 0)
            assert len(f.compute()) == 0
    finally:
        ctx.set_option("distmap_build", 0)


def test_projection_arithmetic_exhaustive_random(ctx, po):
    """Stress the fixed-operation-sequence contract (hand-written divide, polynomial atan2, filtered sqrt): 3 million random
    points, all magnitudes and octants, 16 384 columns -- any single column or depth mismatch changes a winner."""
    rng = np.random.default_rng(123)
    n = 3_000_000
    r = np.exp(rng.uniform(np.log(0.05), np.log(60.0), n)); a = rng.uniform(-np.pi, np.pi, n)
    pts = np.stack([r * np.cos(a), r * np.sin(a), np.cos(a), np.sin(a)], 1).astype(np.float32)
    pts[:1000, 1] = 0.0; pts[1000:2000, 0] = 0.0; pts[2000:2100, :2] = 0.0           # axes and the origin
    pts[2100:2200, 1] = np.float32(1e-30) * pts[2100:2200, 0]                         # subnormal quotients
    pts[2200:2300, 1] = -0.0
    for cols, pose, off in ((16384, [0.0, 0.0, 0.0], 0.0), (16384, [0.3, -0.2, 1.1], 0.0), (4096, [-5.0, 7.0, -2.9], 0.5)):
        pr = api.PointNormal2fProjectorPolar(cols, -math.pi, math.pi, 0.1, 50.0, off)
        src, depth, xyn = pr.compute(ctx, pts, np.float32(pose))
        osrc, odepth, oxyn = po.project(po.Projector(cols, -math.pi, math.pi, 0.1, 50.0, off), pts, np.float32(pose))
        assert (osrc >= 0).sum() > 0.9 * cols
        assert np.array_equal(src, osrc) and np.array_equal(depth, odepth)


def test_non_finite_and_far_away_points_are_inert(ctx, po, small_workload):
    """NaN, +-Inf and absurdly distant points (a corrupted message, an uninitialised buffer) must never become an index: appended to the
    END of a cloud -- so the good points keep their indices -- they change nothing.  Projective paths (finder, aligner, clipper, merger)
    take all of them; the point-query finders take NaN and far-away points (an infinite bounding box is refused with an error, not a fault);
    the preprocessor takes NaN / Inf / negative ranges."""
    wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    nan, inf = np.float32("nan"), np.float32("inf")
    bad_all = np.float32([[nan, 1, 0, 1], [1, nan, 1, 0], [inf, 2, 0, 1], [-inf, inf, 1, 0], [3, -inf, 0, 1], [1e30, -1e30, 1, 0], [nan, nan, nan, nan]])
    bad_fin = np.float32([[nan, 1, 0, 1], [1, nan, 1, 0], [2.5e5, -3e5, 1, 0], [nan, nan, nan, nan]])
    x0 = wl.x0[1]
    for finder, bad in ((api.CorrespondenceFinderProjective2f(ctx, _projector()), bad_all),
                        (api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3), bad_fin),
                        (api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05), bad_fin)):
        for role in ("A", "B"):
            if role == "B" and isinstance(finder, api.CorrespondenceFinderProjective2f):
                continue
            fixed, moving, pose = (scan, wl.map_points, x0) if role == "A" else (wl.map_points, scan, synth.invert_poses(x0[None, :].astype(np.float64))[0].astype(np.float32))
            if isinstance(finder, api.CorrespondenceFinderNN2D):
                bad = bad_fin[[0, 1, 3]]                     # (a 500 km bounding box at 5 cm per pixel is refused: covered below)
            finder.setFixed(fixed); finder.setMoving(moving); finder.setLocalMapInSensor(pose)
            clean = finder.compute()
            finder.setFixed(np.concatenate([fixed, bad])); finder.setMoving(np.concatenate([moving, bad])); finder.setLocalMapInSensor(pose)
            dirty = finder.compute()
            assert len(clean) > 100 and np.array_equal(clean, dirty), (type(finder).__name__, role)
            al = api.MultiAligner2D(ctx, max_iterations=6, min_num_inliers=10)
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10))
            a = al.compute_batch([fixed], [moving], pose[None, :]); b = al.compute_batch([np.concatenate([fixed, bad])], [np.concatenate([moving, bad])], pose[None, :])
            assert a.status[0] == 0 and np.array_equal(a.pose, b.pose) and np.array_equal(a.information, b.information), (type(finder).__name__, role)
    f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.05)
    f.setFixed(np.concatenate([scan, bad_all])); f.setMoving(wl.map_points); f.setLocalMapInSensor(x0)
    with pytest.raises(api.Lsm2dError):
        f.compute()                                          # infinite bounding box: an error, and the context stays usable
    # clipper / merger
    robot = synth.invert_poses(wl.x_true[1:2])[0].astype(np.float32)
    clip = api.SceneClipperProjective2D(ctx, _projector(), voxelize_resolution=0.0)
    clip.setFullScene(wl.map_points); clip.setRobotInLocalMap(robot); c0 = clip.compute().download()
    clip.setFullScene(np.concatenate([wl.map_points, bad_all])); c1 = clip.compute().download()
    assert len(c0) > 300 and np.array_equal(c0, c1)
    scene = api.CloudSet.reserved(ctx, 40000); scene.upload(np.concatenate([wl.map_points, bad_all]))
    mg = api.MergerProjective2D(ctx, _projector(), 0.2); mg.setScene(scene); mg.setMeasurement(np.concatenate([scan, bad_all])); mg.setMeasurementInScene(robot)
    mg.compute()
    want, _ = po.merge_scene(po.Projector(1081, -math.pi, math.pi, 0.3, 30.0, 0.0), wl.map_points, scan, robot, 0.2)
    got = scene.download()
    keep = np.ones(len(got), bool); keep[len(wl.map_points):len(wl.map_points) + len(bad_all)] = False       # the bad scene points stay where they were, untouched
    assert np.array_equal(got[keep], want) and np.array_equal(got[~keep], bad_all, equal_nan=True)
    # preprocessor
    world = synth.make_world(2); a0, a1 = -2.34747, 2.35619
    r = synth.make_scan_ranges(world, synth.sample_poses(world, 2, seed=4), n_beams=721, angle_min=a0, angle_max=a1, noise_sigma=0.005, seed=1)
    r[0, 10:20] = nan; r[0, 100] = inf; r[0, 200:205] = -1.0; r[1, :] = nan
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    pre.setRawData(r, a0, a1, 0.0, 30.0); cs = pre.compute()
    pp = po.Preprocessor(721, a0, a1, 0.3, 20.0, 0.3, 5, 0.02)
    assert np.array_equal(cs.download(0), po.preprocess_scan(pp, r[0])) and cs.counts[1] == 0 and np.isfinite(cs.download(0)).all()


def test_nn_cooperative_search_is_chosen_per_alignment(ctx, po, small_workload):
    """A ragged NN batch: alignment 0 searches a fixed cloud more than four times its moving one (four lanes per query), alignment 1 a
    fixed cloud smaller than that (one lane per query).  The loop is picked per alignment from the device-side counts; both must
    carry the device-order mirror's bits (the mirror applies the same rule), i.e. the summation order follows the loop actually run."""
    wl = small_workload
    scan0 = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]; scan1 = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    big = wl.map_points; small = wl.map_points[::12]                      # ~30000 vs ~2500 fixed points; the scans have ~1000
    assert len(big) >= 4 * len(scan0) and len(small) < 4 * len(scan1)
    fixed = api.CloudSet(ctx, np.concatenate([big, small], 0), np.array([0, len(big), len(big) + len(small)], np.int32))
    moving = api.CloudSet(ctx, np.concatenate([scan0, scan1], 0), np.array([0, len(scan0), len(scan0) + len(scan1)], np.int32))
    x0 = synth.invert_poses(wl.x0[:2].astype(np.float64)).astype(np.float32)       # scan-in-map estimates
    al = api.MultiAligner2D(ctx, max_iterations=12, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4, normal_cos=0.8), min_num_correspondences=10))
    res = al.compute_batch([fixed], [moving], x0, want_stats=True)
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=0.4)
    for i, (f, m) in enumerate(((big, scan0), (small, scan1))):
        rt = po.align(po.aligner_params(12, device_order=True), [osp], [f], [m], x0[i])
        _assert_bitwise_equal_to_device_order_oracle(res, i, rt, ("ragged nn", i))
    assert res.status[0] == 0


@pytest.mark.parametrize("n_map", [10000, 100000, 1000000])
def test_kdtree_finder_bit_exact_both_roles(ctx, po, n_map):
    """KDTree2D(coordinates, max_leaf_range, min_leaf_points) built on the device + findNeighbor's single-leaf descent: the SAME pairs, in
    the same order, as the oracle's restatement of the believed upstream tree (lsmo_find_kdtree_f -- unchanged by this build: sequential
    sums, unfused products) in both roles at the three map sizes of BASELINE.json, for the class defaults and a second parameter set, and
    with both forms of the build's sequential sums (systolic DPP pass / one v_readlane per value)."""
    wl = synth.make_workload(3, n_map, seed=11)
    scan = wl.scan_points[wl.scan_offsets[1]:wl.scan_offsets[2]]
    x = wl.x0[1]
    xb = synth.invert_poses(x[None, :].astype(np.float64))[0].astype(np.float32)
    for leaf_range, leaf_points in ((1e-2, 20), (0.05, 7)):
        osp = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5, normal_cos=0.8, kd_max_leaf_range=leaf_range, kd_min_leaf_points=leaf_points)
        for role, (fixed, moving, pose) in enumerate(((scan, wl.map_points, x), (wl.map_points, scan, xb))):     # role A (tracker wiring), role B (BASELINE wording)
            want = po.find(osp, fixed, moving, pose)
            assert len(want) > 300
            # (round 4: a workgroup per node on the top levels of a map-sized cloud, or a wave per node throughout; the first pair is what the library ships with,
            # the others are forced through knobs of the experiments build)
            for chain, wide in ((1, 1024), (1, 4096), (0, 4096), (1, 0)):
                try:
                    if not xset(ctx, kd_chain=chain, kd_wide_min_points=wide):
                        continue
                    f = _kd_finder(ctx, 0.5, leaf_range, leaf_points)
                    f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
                    got = f.compute()
                finally:
                    xset(ctx, kd_chain=1, kd_wide_min_points=1024)
                assert np.array_equal(got, want), (n_map, role, leaf_range, chain, wide, len(got), len(want))
            # the tree is approximate by construction: it must NOT be the exact search (else this test would not tell the two apart)
            ex = po.find(po.slice_params(finder=po.FINDER_NN, max_distance=0.5, normal_cos=0.8), fixed, moving, pose)
            assert not np.array_equal(ex, want)


def test_kdtree_finder_edge_cases(ctx, po):
    """Degenerate trees: empty / one-point / two-point clouds, fewer points than min_leaf_points (the root is a leaf), all points identical
    (no axis), collinear and duplicate points (ties -> the lowest index of the leaf), a leaf range that makes the root a leaf, min_leaf_points
    of 1 and 2 (the deepest trees), queries far outside the cloud."""
    rng = np.random.default_rng(9)
    def cloud(n, lo=-5, hi=5):
        p = rng.uniform(lo, hi, size=(n, 2)); a = rng.uniform(-np.pi, np.pi, n)
        return np.concatenate([p, np.cos(a)[:, None], np.sin(a)[:, None]], 1).astype(np.float32)
    moving = cloud(700, -6, 6)
    pose = np.float32([0.1, -0.2, 0.3])
    cases = []
    for n in (0, 1, 2, 3, 19, 20, 21, 64, 65, 129, 1000):
        cases.append((cloud(n), 1e-2, 20))
    ident = np.tile(cloud(1), (300, 1)); cases.append((ident, 1e-2, 20))
    line = cloud(500); line[:, 1] = np.float32(0.25); cases.append((line, 1e-2, 20))
    dup = cloud(400); dup[100:200] = dup[0:100]; cases.append((dup, 1e-3, 2))
    cases.append((cloud(3000), 100.0, 20))             # extent below max_leaf_range at once: one leaf holding everything
    cases.append((cloud(3000), 1e-3, 1)); cases.append((cloud(3000), 1e-3, 2)); cases.append((cloud(5000), 0.3, 50))
    grid = np.stack(np.meshgrid(np.arange(40), np.arange(40)), -1).reshape(-1, 2).astype(np.float32) * 0.25        # exact ties in the covariance
    cases.append((np.concatenate([grid, np.tile(np.float32([1, 0]), (len(grid), 1))], 1), 0.2, 4))
    for k, (fixed, lr, lp) in enumerate(cases):
        for md in (0.05, 0.4, 3.0):
            f = _kd_finder(ctx, md, lr, lp, normal_cos=-1.0)
            f.setFixed(fixed) if len(fixed) else f.setFixed(np.zeros((0, 4), np.float32))
            f.setMoving(moving); f.setLocalMapInSensor(pose)
            got = f.compute()
            want = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=md, normal_cos=-1.0, kd_max_leaf_range=lr, kd_min_leaf_points=lp),
                           fixed if len(fixed) else np.zeros((0, 4), np.float32), moving, pose)
            assert np.array_equal(got, want), (k, len(fixed), lr, lp, md, len(got), len(want))
    # the class defaults apply when the parameters are not set (<= 0), as in the oracle
    f = _kd_finder(ctx, 0.4, 0.0, 0, normal_cos=-1.0); fixed = cloud(4000)
    f.setFixed(fixed); f.setMoving(moving); f.setLocalMapInSensor(pose)
    assert np.array_equal(f.compute(), po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.4, normal_cos=-1.0), fixed, moving, pose))


def test_two_kdtree_slices_build_their_scans_trees_in_one_launch(ctx, po):
    """The live tracker with the reference's KD-tree finder: one alignment, two slices, each with its own NEW scan as the fixed cloud (a tree per scan and
    step: CorrespondenceFinderKDTree2D::reset, correspondence_finder_kd_tree_2d.cpp:6-8,31-38), the same scene as the moving cloud.  The aligner call
    builds both trees side by side in one launch (k_kd_build_scan_multi) -- the same poses, information matrices and statistics as with each tree built by
    the workgroup build of its own call ("kd_scan_max_clouds" 0), step after step with refilled reserved sets, and the oracle's bits."""
    wl = synth.make_workload(8, 20000, seed=15)
    scene = wl.map_points[::25][:700].copy()
    m0 = api.CloudSet.reserved(ctx, 1400); m1 = api.CloudSet.reserved(ctx, 1400); sc = api.CloudSet(ctx, scene)
    f0 = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.8, max_leaf_range=0.01, min_leaf_points=20, search="kdtree")
    f1 = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.25, normal_cos=0.7, max_leaf_range=0.02, min_leaf_points=12, search="kdtree")
    al = api.MultiAligner2D(ctx, max_iterations=8, min_num_inliers=5)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(f0, min_num_correspondences=5))
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(f1, min_num_correspondences=5, robustifier=api.RobustifierCauchy(0.05)))
    osp = [po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3, normal_cos=0.8, kd_max_leaf_range=0.01, kd_min_leaf_points=20, min_num_correspondences=5),
           po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.25, normal_cos=0.7, kd_max_leaf_range=0.02, kd_min_leaf_points=12, min_num_correspondences=5,
                           robustifier=po.ROBUST_CAUCHY, chi_threshold=0.05)]
    try:
        for step in range(4):
            a = wl.scan_points[wl.scan_offsets[2 * step]:wl.scan_offsets[2 * step + 1]]
            b2 = wl.scan_points[wl.scan_offsets[2 * step + 1]:wl.scan_offsets[2 * step + 2]]
            x0 = wl.x0[2 * step][None, :]
            got = {}
            for scan_max in (8, 0):      # (0: each tree by the workgroup build of its own call -- a knob of the experiments build)
                if not xset(ctx, kd_scan_max_clouds=scan_max):
                    continue
                m0.upload(a); m1.upload(b2)           # new scans: both trees are rebuilt
                got[scan_max] = al.compute_batch([m0, m1], [sc, sc], x0, want_stats=True)
            g, h = got[8], got.get(0, got[8])
            assert np.array_equal(g.pose, h.pose) and np.array_equal(g.information, h.information) and np.array_equal(g.status, h.status) and np.array_equal(g.stats, h.stats), step
            rt = po.align(po.aligner_params(8, min_num_inliers=5, device_order=True), osp, [a, b2], [scene, scene], x0[0])
            _assert_bitwise_equal_to_device_order_oracle(g, 0, rt, ("two kd slices", step))
    finally:
        xset(ctx, kd_scan_max_clouds=8)


@pytest.mark.gpu
def test_grid_nn_over_the_map_position_search_ties_and_cell_cache(ctx, po):
    """The grid NN with a map-sized fixed cloud runs in an instantiation of its own (k_align<0,1,0,0,1>): the search keeps the winner's position,
    reads an original index only to break an exact tie, and caches every query's cell ranges in LDS between iterations.  A map with DUPLICATED
    points (exact ties of distances on most queries: the lower original index must win, as in the oracle) aligned with the cache on and off,
    ragged scans, Cauchy: the same bits both ways, and the device-order oracle's."""
    wl = synth.make_workload(12, 40000, seed=12)
    dup = np.concatenate([wl.map_points, wl.map_points[::3], wl.map_points[5000:9000]], 0)      # every third point twice, a stretch three times
    x0_b = synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32)
    scans = [wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]][: 1081 - 37 * i] for i in range(12)]
    offs = np.concatenate([[0], np.cumsum([len(s) for s in scans])]).astype(np.int32)
    fixed = api.CloudSet(ctx, dup); moving = api.CloudSet(ctx, np.concatenate(scans, 0), offs)
    al = api.MultiAligner2D(ctx, max_iterations=12, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4, normal_cos=0.7, search="exact"),
                                                                      min_num_correspondences=10, robustifier=api.RobustifierCauchy(0.03)))
    res = {}
    for cache in (1, 0):      # (0: without the per-query cell cache -- a knob of the experiments build)
        try:
            if xset(ctx, nn_qcache=cache):
                res[cache] = al.compute_batch([fixed], [moving], x0_b, want_stats=True)
        finally:
            xset(ctx, nn_qcache=1)
    a, c = res[1], res.get(0, res[1])
    assert np.array_equal(a.pose, c.pose) and np.array_equal(a.information, c.information) and np.array_equal(a.status, c.status) and np.array_equal(a.stats, c.stats)
    assert np.all(a.status == 0)
    osp = po.slice_params(finder=po.FINDER_NN, max_distance=0.4, normal_cos=0.7, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.03, min_num_correspondences=10)
    for i in (0, 5, 11):
        rt = po.align(po.aligner_params(12, min_num_inliers=10, device_order=True), [osp], [dup], [scans[i]], x0_b[i])
        _assert_bitwise_equal_to_device_order_oracle(a, i, rt, ("grid NN over the map", i))
    # the finder-level call on the same clouds returns the oracle's pairs (lowest index on every tie)
    f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.4, normal_cos=0.7, search="exact")
    f.setFixed(dup); f.setMoving(scans[0]); f.setLocalMapInSensor(x0_b[0])
    got = f.compute(); want = po.find(po.slice_params(finder=po.FINDER_NN, max_distance=0.4, normal_cos=0.7), dup, scans[0], x0_b[0])
    assert np.array_equal(got, want) and len(want) > 500
    tied = np.isin(want[:, 0], np.arange(0, len(wl.map_points), 3)).mean()
    assert tied > 0.2, tied          # many winners ARE the lower-indexed copy of a duplicated point


def test_kdtree_single_launch_build_equals_the_level_loop(ctx, po):
    """Round 4: scan-sized clouds get their KD-tree from ONE launch (k_kd_build_wg: a workgroup per cloud walks the levels itself -- the reference
    rebuilds the tree whenever the fixed cloud changes, correspondence_finder_kd_tree_2d.cpp:6-8,31-38, i.e. per scan in the live tracker); the
    level-by-level build of round 3 stays for map-sized clouds ("kd_wg_max_points" 0 forces it).  Same kd_node, same order of every sequential sum:
    the same trees -- node counts, depths, and every pair of every query -- and both equal the oracle's."""
    world = synth.make_world(12)
    robots = synth.sample_poses(world, 24, seed=3)
    pts, offs = synth.make_scans(world, robots, n_beams=1081, noise_sigma=0.004, seed=2)
    m = synth.make_map(world, 12000, noise_sigma=0.002, seed=4)
    degenerate = [np.zeros((0, 4), np.float32), pts[:1], pts[:2], pts[:19], pts[:20], pts[:21], np.repeat(pts[:1], 50, 0)]
    clouds = [pts[offs[i]:offs[i + 1]] for i in range(24)] + degenerate + [m]
    offs_all = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int32)
    allp = np.concatenate(clouds, 0)
    x0 = np.float32([0.02, -0.01, 0.01])
    res = {}
    try:
        # 100: the scans go through the level loop, the tiny clouds through the workgroup build (a mixed set); the level loop with a WORKGROUP per node
        # (kd_node_wide: "kd_wide_min_points", by default only the top levels of a map-sized cloud) on every level that holds 64 / 1000 points per node
        # (the first pair is what the library ships with; the others force the other forms of the build through knobs of the experiments build)
        for wg, wide in ((16384, 1024), (16384, 4096), (0, 0), (100, 0), (0, 64), (100, 1000)):
            if not xset(ctx, kd_wg_max_points=wg, kd_wide_min_points=wide):
                continue
            cs = api.CloudSet(ctx, allp, offs_all)
            out = []
            for lr, lp in ((1e-2, 20), (0.05, 7)):
                f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.5, max_leaf_range=lr, min_leaf_points=lp, search="kdtree")
                for ci in range(len(clouds)):
                    f.setFixed(cs, ci); f.setMoving(m[::7]); f.setLocalMapInSensor(x0)
                    out.append(f.compute())
                out.append(np.array([[ctx.get_option("last_kd_levels"), ctx.get_option("last_kd_nodes")]]))
            res[(wg, wide)] = out
            cs.close()
    finally:
        xset(ctx, kd_wg_max_points=16384, kd_wide_min_points=1024)
    for key in res:
        assert len(res[key]) == len(res[(16384, 1024)])
        for a, b in zip(res[(16384, 1024)], res[key]):
            assert np.array_equal(a, b), key
    k = 0
    for lr, lp in ((1e-2, 20), (0.05, 7)):
        for ci in (0, 5, 23, 24, 27, 29, len(clouds) - 1):
            want = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3, normal_cos=0.5, kd_max_leaf_range=lr, kd_min_leaf_points=lp), clouds[ci], m[::7], x0)
            assert np.array_equal(res[(16384, 1024)][k + ci], want), (lr, lp, ci)
        k += len(clouds) + 1
    # the LATENCY form (k_kd_build_scan: a set of at most "kd_scan_max_clouds" clouds of <= 1280 points, working set in LDS, sixteen waves, groups of four
    # waves on the levels with few nodes): one scan; eight clouds with the degenerate ones among them; both forms of the chains -- against the workgroup
    # build ("kd_scan_max_clouds" 0) and the oracle
    small_sets = ([clouds[0]], [clouds[3][:1280]], [clouds[1], clouds[2]] + degenerate[:6], [degenerate[6], clouds[7][:700], clouds[8][:65], clouds[9][:64], clouds[10][:129]])
    try:
        for cl in small_sets:
            o = np.concatenate([[0], np.cumsum([len(q) for q in cl])]).astype(np.int32)
            ap_ = np.concatenate(cl, 0)
            got = {}
            for scan_max, chain in ((8, 1), (8, 0), (0, 1)):
                if not xset(ctx, kd_scan_max_clouds=scan_max, kd_chain=chain):
                    continue
                cs = api.CloudSet(ctx, ap_, o)
                out = []
                for lr, lp in ((1e-2, 20), (0.05, 7)):
                    f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.5, max_leaf_range=lr, min_leaf_points=lp, search="kdtree")
                    for ci in range(len(cl)):
                        f.setFixed(cs, ci); f.setMoving(m[::7]); f.setLocalMapInSensor(x0)
                        out.append(f.compute())
                    out.append(np.array([[ctx.get_option("last_kd_levels"), ctx.get_option("last_kd_nodes")]]))
                got[(scan_max, chain)] = out
                cs.close()
            for key in got:
                for a, b in zip(got[(8, 1)], got[key]):
                    assert np.array_equal(a, b), (key, [len(q) for q in cl])
            k = 0
            for lr, lp in ((1e-2, 20), (0.05, 7)):
                for ci in range(len(cl)):
                    want = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3, normal_cos=0.5, kd_max_leaf_range=lr, kd_min_leaf_points=lp), cl[ci], m[::7], x0)
                    assert np.array_equal(got[(8, 1)][k + ci], want), (lr, lp, ci, len(cl[ci]))
                k += len(cl) + 1
    finally:
        xset(ctx, kd_scan_max_clouds=8, kd_chain=1)
