"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bars (task statement / BASELINE.json north_star):
  * index work -- z-buffer winners, correspondence pairs -- BIT-EXACT against the fp32 oracle;
  * H / b / chi: relative 2e-5 of the term magnitude against the fp64 oracle (fp32 tree vs sequential sums);
  * aligner pose: within POSE_TOL_M = 1e-4 m / POSE_TOL_RAD = 1e-4 rad of the oracle (fp32 and fp64) and,
    on noise-free data, of the generating pose.
"""
import json
import math

import numpy as np
import pytest

import fuzz_cases
from conftest import golden_path, has_experiments, need_experiments, xset
from srrg2_laser_slam_2d_amd import api, synth

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4


_oracle_slice = fuzz_cases.oracle_slice      # oracle SliceParams with the same values as an ABI SliceParams


def _same_correspondence_sets(gpu_stats, oracle_stats, iterations):
    """Did the device and the fp32 oracle use the SAME pairs in every iteration?  Decided EXACTLY since round 4: every iteration's statistics
    carry an order-independent 64-bit digest of its correspondence set (lsm2d_iteration_stats.pair_digest: the wrapping sum of a hash of
    (slice, fixed index, moving index) over the pairs), formed by the kernels and by the oracle alike -- the pairs are an observable of the
    reference's aligner (apps/visual_test_aligner_2d.cpp:129-143).  (Rounds 2-3 inferred it from counts and chi^2 sums: equal counts are
    necessary, not sufficient, and equal sums to 3e-4 did not prove equal pairs either.)"""
    dg = api.pair_digests(gpu_stats[:iterations])
    for k in range(iterations):
        g, o = gpu_stats[k], oracle_stats[k]
        if int(g["n_correspondences"]) != o.n_corr or int(dg[k]) != o.pair_digest:
            return False
    return True


def _assert_bitwise_equal_to_device_order_oracle(res, i, rt, tag):
    """The fp32 oracle with lsmo_aligner_params.device_order = 1 sums in the kernels' order: everything must be equal BITWISE --
    status, iteration count, pose, information matrix, and every iteration's counts and chi^2 sums (inliers and kernelised outliers)."""
    assert int(res.status[i]) == rt["status"] and int(res.iterations[i]) == rt["iterations"], (tag, res.status[i], rt["status"], res.iterations[i], rt["iterations"])
    assert np.array_equal(res.pose[i], rt["pose"]), (tag, "pose", res.pose[i].tolist(), rt["pose"].tolist())
    assert np.array_equal(res.information[i], rt["H"]), (tag, "H", res.information[i].tolist(), rt["H"].tolist())
    if res.stats is not None:
        for k in range(rt["iterations"]):
            g, o = res.stats[i][k], rt["stats"][k]
            assert (int(g["n_correspondences"]), int(g["n_inliers"]), int(g["n_outliers"])) == (o.n_corr, o.n_in, o.n_out), (tag, "counts", k)
            assert np.float32(g["chi_inliers"]) == np.float32(o.chi_in), (tag, "chi_in", k, float(g["chi_inliers"]), o.chi_in)
            assert np.float32(g["chi_outliers"]) == np.float32(o.chi_out), (tag, "chi_out", k, float(g["chi_outliers"]), o.chi_out)
            assert (int(g["pair_digest_hi"]) << 32 | int(g["pair_digest_lo"])) == o.pair_digest, (tag, "pair digest", k)      # the same correspondence SET, exactly


def _pose_diff(p, q):
    d = np.abs(np.asarray(p, np.float64) - np.asarray(q, np.float64)); d[2] = abs((d[2] + math.pi) % (2 * math.pi) - math.pi)
    return float(d[:2].max()), float(d[2])


class _Envelope:
    """Round 5: what the fuzz tests hold an alignment to when it is NOT in the strict class (every iteration's pair digest equal to the
    sequential fp32 oracle's AND that oracle within 2.5e-5 of the fp64 one: bar 1e-4 m / 1e-4 rad against the fp32 oracle).  The device -- which equals the
    device-order mirror bit for bit in every alignment anyway -- may then be as far from the fp64 TRUTH as the reference's own arithmetic is, not a flat
    centimetre: |device - fp64| <= max(1e-4, 3 x max(|sequential fp32 - fp64|, |reference-arithmetic fp32 (_r: libm, no FMA) - fp64|)), metres and radians
    separately.  Both of those fp32 evaluations sum pair after pair; the device sums in trees, and where a pair sits on a gate the two orders pick different
    pairs.  Whether an alignment is SENSITIVE to that is a property of the problem, and the reference's own arithmetic shows it: the sequential fp32 oracle is
    run again from the start pose moved by ONE ULP per component (four sign patterns) -- what another compiler's last bit would do to the reference -- and the
    envelope takes those runs in.  An alignment that only passes with them is tallied apart (needs_perturbed).
    Tallies: ok (within 1e-4 of fp64 outright), needs_factor (within 3 x the two evaluations' distance), needs_perturbed (within 3 x the distance of the
    one-ulp-perturbed runs), no_oracle (the fp64 oracle, or every fp32 one, did not succeed: nothing to compare with), status_differs (fp64 and an fp32 oracle
    succeed, the device's mirror does not), ill_conditioned (the REFERENCE arithmetic has no answer: its own evaluations, or its runs from one-ulp-moved start poses,
    end more than 1e-2 m / 1e-2 rad -- a hundred bars -- from the fp64 oracle: an alignment that diverges chaotically (seed 4711 / trial 219 of the parameter fuzz:
    one ulp on the start pose moves the sequential oracle by 0.74 m, the device's tree sums by 5 m).  Three times a four-sample spread bounds nothing there; such
    alignments are counted, listed and bounded in number, and keep the bitwise device-order check like every other one), violation (outside all of it)."""
    ILL = 1e-2

    def __init__(self, test="", seed=0):
        self.test, self.seed = test, seed      # what a named exception is looked up by (fuzz_cases.KNOWN_TREE_ORDER_DEVIATIONS / KNOWN_ILL_CONDITIONED)
        self.tally = dict(ok=0, needs_factor=0, needs_perturbed=0, no_oracle=0, status_differs=0, ill_conditioned=0, violation=0)
        self.violations = []
        self.ill = []
        self.worst = dict(ok=0.0, needs_factor=0.0, needs_perturbed=0.0)

    @staticmethod
    def one_ulp_starts(x0):
        """Round 6: 64 starts (all 26 one-ulp patterns, then two-ulp ones: fuzz_cases.perturbed_starts) instead of four -- the replay of round 5's five violators
        (profiles/r06/violators_replay_r06.txt) showed the reference's own arithmetic leaving 1e-4 in 10-12 of 64 such runs for two of them and reaching 0.95e-4
        for a third, where the four-run sample had seen nothing"""
        return [x for _, x in fuzz_cases.perturbed_starts(x0, 64)]

    def assert_only_named_exceptions(self):
        """Round 6: no allowance by COUNT any more.  Outside the envelope may lie only the alignments NAMED in fuzz_cases -- the two of the
        eighteen-seed soak where the tree order alone lands a pair on the other side of a gate, each with its own bound -- and ill-conditioned may be only the one
        named there.  The default run (seeds 5 / 2024) holds none of them: zero tolerated."""
        import os
        collect = bool(os.environ.get("LSM2D_FUZZ_COLLECT"))      # a soak that LISTS what is not named yet instead of stopping at the first (tools/fuzz_soak.sh collect)
        for where, v in self.violations:      # where = (trial, alignment, note)
            key = (self.test, int(self.seed), int(where[0]), int(where[1]))
            bound = fuzz_cases.KNOWN_TREE_ORDER_DEVIATIONS.get(key)
            if collect and (bound is None or max(v["device_vs_fp64"]) > bound):
                print("UNNAMED OUTSIDE", key, v); continue
            assert bound is not None, ("outside the envelope and not a named tree-order deviation", key, v)
            assert max(v["device_vs_fp64"]) <= bound, ("a named tree-order deviation beyond its recorded bound", key, v)
        for where, v in self.ill:
            key = (self.test, int(self.seed), int(where[0]), int(where[1]))
            if collect and key not in fuzz_cases.KNOWN_ILL_CONDITIONED:
                print("UNNAMED ILL-CONDITIONED", key, v); continue
            assert key in fuzz_cases.KNOWN_ILL_CONDITIONED, ("ill-conditioned and not a named case", key, v)

    def check(self, where, dev_pose, dev_status, r, rd, rr, perturbed=None):
        """perturbed: callable -> the sequential fp32 oracle's results from one_ulp_starts(x0); asked for only when the two evaluations' envelope does not hold"""
        if rd["status"] != 0:
            self.tally["no_oracle"] += 1; return "no_oracle"
        oracles = [o for o in (r, rr) if o["status"] == 0]
        if not oracles:
            self.tally["no_oracle"] += 1; return "no_oracle"
        if dev_status != 0:
            self.tally["status_differs"] += 1; return "status_differs"
        em = max(_pose_diff(o["pose"], rd["pose"])[0] for o in oracles); er = max(_pose_diff(o["pose"], rd["pose"])[1] for o in oracles)
        dm, dr = _pose_diff(dev_pose, rd["pose"])
        if dm <= POSE_TOL_M and dr <= POSE_TOL_RAD:
            self.tally["ok"] += 1; self.worst["ok"] = max(self.worst["ok"], dm, dr); return "ok"
        if dm <= max(POSE_TOL_M, 3.0 * em) and dr <= max(POSE_TOL_RAD, 3.0 * er):
            self.tally["needs_factor"] += 1; self.worst["needs_factor"] = max(self.worst["needs_factor"], dm, dr); return "needs_factor"
        pm = pr_ = 0.0
        if perturbed is not None:
            more = [o for o in perturbed() if o["status"] == 0]
            if more:
                pm = max(_pose_diff(o["pose"], rd["pose"])[0] for o in more); pr_ = max(_pose_diff(o["pose"], rd["pose"])[1] for o in more)
        # (ill-conditioned is decided BEFORE the perturbed runs are allowed to widen the envelope: with 64 of them, three times a spread of metres would cover anything)
        if max(em, pm) > self.ILL or max(er, pr_) > self.ILL:
            self.tally["ill_conditioned"] += 1
            self.ill.append((where, dict(device_vs_fp64=(dm, dr), oracles_vs_fp64=(em, er), one_ulp_runs_vs_fp64=(pm, pr_))))
            return "ill_conditioned"
        if perturbed is not None and dm <= max(POSE_TOL_M, 3.0 * em, 3.0 * pm) and dr <= max(POSE_TOL_RAD, 3.0 * er, 3.0 * pr_):
            self.tally["needs_perturbed"] += 1; self.worst["needs_perturbed"] = max(self.worst["needs_perturbed"], dm, dr); return "needs_perturbed"
        self.tally["violation"] += 1
        self.violations.append((where, dict(device_vs_fp64=(dm, dr), oracles_vs_fp64=(em, er), one_ulp_runs_vs_fp64=(pm, pr_), device=np.asarray(dev_pose).tolist(), fp64=np.asarray(rd["pose"]).tolist())))
        return "violation"

    def summary(self):
        t = self.tally
        return ("envelope class: %d within 1e-4 of the fp64 oracle outright (worst %.2e), %d within 3 x the reference arithmetic's own distance from it (worst %.2e), "
                "%d within 3 x what ONE ULP on the start pose does to the reference arithmetic (worst %.2e), %d with no oracle to compare with, %d where only the "
                "device-order evaluation fails, %d ill-conditioned (the reference arithmetic itself spreads over more than 1e-2), %d OUTSIDE the envelope"
                % (t["ok"], self.worst["ok"], t["needs_factor"], self.worst["needs_factor"], t["needs_perturbed"], self.worst["needs_perturbed"], t["no_oracle"], t["status_differs"],
                   t["ill_conditioned"], t["violation"]))


def _projector(cols=1081, rmin=0.3, rmax=30.0, off=0.0):
    return api.PointNormal2fProjectorPolar(cols, -math.pi, math.pi, rmin, rmax, off)


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


def _aligner(ctx, cols=1081, its=20, **slice_kw):
    finder = api.CorrespondenceFinderProjective2f(ctx, _projector(cols))
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10, **slice_kw))
    return al


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


def test_full_size_batch_properties(ctx):
    """BASELINE configs[1] at full size: 1000 scans x 100k-point map x 20 iterations.  Too slow for the
    scalar oracle in a unit test, so it is checked through size-independent properties: convergence to the
    generating pose on noise-free data, run-to-run bitwise determinism, permutation equivariance."""
    wl = synth.make_workload(1000, 100000, seed=0)
    al = _aligner(ctx)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    res = al.compute_batch([fixed], [moving], wl.x0)
    assert np.all(res.status == 0)
    d = np.abs(res.pose - wl.x_true)
    assert d[:, :2].max() < POSE_TOL_M and d[:, 2].max() < POSE_TOL_RAD
    perm = np.argsort(synth.Stream(9).uniform(1000)).astype(np.int32)
    res_p = al.compute_batch([fixed], [moving], wl.x0[perm], fixed_index=perm[None, :])
    assert np.array_equal(res_p.pose, res.pose[perm])


def test_device_resident_input(ctx, small_workload):
    import torch
    wl = small_workload
    t = torch.from_numpy(wl.map_points).cuda()
    a = api.CloudSet(ctx, t); b = api.CloudSet(ctx, wl.map_points)
    pr = _projector()
    sa = pr.compute(ctx, a, wl.x0[0]); sb = pr.compute(ctx, b, wl.x0[0])
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[1], sb[1])


def test_srrg_adapters_compile_and_run(ctx, po, small_workload, tmp_path):
    """The SRRG-side adapter sources (adapters/srrg/*: three finder siblings, MultiAlignerHIP2D, clipper / merger / raw-data preprocessor siblings) compiled
    against the stand-in srrg2 headers of tests/cpp/adapter_shim, linked with the real library and driven as the reference drives its own
    classes (tests/cpp/adapter_driver.cpp): same pairs as the oracle, the aligner's pose / status / statistics written back, the
    odometry-prior slice translated, an in-place change of the moving cloud seen, an unknown slice processor refused."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "adapter_driver")
    lib_dir = os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "lib"); ad = os.path.join(ROOT, "adapters", "srrg")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + ad, "-I" + os.path.join(ROOT, "tests", "cpp", "adapter_shim"),
                    os.path.join(ROOT, "tests", "cpp", "adapter_driver.cpp"), os.path.join(ad, "correspondence_finder_hip_2d.cpp"),
                    os.path.join(ad, "multi_aligner_hip_2d.cpp"), "-L" + lib_dir, "-llsm2d_hip", "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    wl = small_workload
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    f.tofile(tmp_path / "fixed.bin"); wl.map_points.tofile(tmp_path / "moving.bin")
    x0 = wl.x0[0]; iters = 12
    world = synth.make_world(2); a0, a1 = -2.34747, 2.35619
    ranges = synth.make_scan_ranges(world, synth.sample_poses(world, 1, seed=4), n_beams=721, angle_min=a0, angle_max=a1, noise_sigma=0.005, seed=1)[0]
    ranges[100:110] = np.inf; ranges.tofile(tmp_path / "ranges.bin")
    out = subprocess.run([exe, str(tmp_path / "fixed.bin"), str(tmp_path / "moving.bin"), repr(float(x0[0])), repr(float(x0[1])), repr(float(x0[2])), "1081", str(iters),
                          str(tmp_path / "ranges.bin"), repr(a0), repr(a1), str(tmp_path / "prep.bin")],
                         check=True, capture_output=True, text=True, timeout=180).stdout
    r = json.loads(out.strip().splitlines()[-1])
    # finders: the pose reaches the ABI as t2v(v2t(x0)) (one atan2 / cos / sin round trip of the stand-in geometry): pairs may differ from the
    # oracle's at x0 by a column or two, not more
    want = po.find(po.slice_params(), f, wl.map_points, x0)
    got = np.array(r["pairs_projective"], np.int32).reshape(-1, 2)
    sa = {tuple(p) for p in want.tolist()}; sb = {tuple(p) for p in got.tolist()}
    assert r["threw_on_missing_inputs"] == 1 and len(sa ^ sb) <= 0.01 * len(sa) and len(sb) > 500
    assert r["in_place_change_seen"] == 1 and r["pairs_before_change"] == len(got) and r["pairs_after_change"] != r["pairs_before_change"]
    # plugin interface #1 under the reference's own aligner loop (round 5): twenty compute() calls over an unchanged 100k-point moving cloud upload it ZERO more
    # times after the first call (content check), with the pairs of a finder that uploads every call; the siblings share one device context unless told otherwise
    assert r["aligner_loop_moving_uploads"] == 0 and r["every_call_uploads"] >= 20 and r["aligner_loop_same_pairs"] == 1 and r["aligner_loop_pairs_last"] > 500
    assert r["siblings_share_a_context"] == 1 and r["own_context_is_separate"] == 1
    assert abs(r["n_kdtree"] - len(po.find(po.slice_params(finder=po.FINDER_NN, max_distance=0.3), f, wl.map_points, x0))) <= 5
    # the KD-tree sibling's default search is the reference's own tree, with the leaf parameters of the configuration; an unknown search is refused
    want_t = po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3, kd_max_leaf_range=0.05, kd_min_leaf_points=12), f, wl.map_points, x0)
    got_t = np.array(r["pairs_kdtree_tree"], np.int32).reshape(-1, 2)
    st = {tuple(p) for p in want_t.tolist()}; sg = {tuple(p) for p in got_t.tolist()}
    assert r["n_kdtree_tree"] == len(got_t) > 100 and len(st ^ sg) <= 0.01 * len(st) and r["threw_on_bad_search"] == 1
    assert abs(r["n_nn"] - len(po.find(po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.1), f, wl.map_points, x0))) <= 5
    # aligner: pose, status enum (stand-in: Success = 3, NotEnoughInliers = 2), iteration statistics, information matrix, slice binding
    o = po.align(po.aligner_params(iters), [po.slice_params()], [f], [wl.map_points], x0)
    d = np.abs(np.array(r["pose"]) - o["pose"])
    assert d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    assert r["status"] == 3 and r["device_status"] == 0 and r["iterations"] == iters and r["slice_fixed_bound"] == 1
    assert abs(r["last_inliers"] - o["stats"][-1].n_in) <= 3 and abs(r["H00"] - o["H"][0, 0]) < 1e-2 * o["H"][0, 0] and r["H22"] > 0
    assert abs(r["slice_pairs"] - o["stats"][-1].n_corr) <= 3
    assert r["pose_again"] == r["pose"]                           # reused device clouds, same bits
    assert r["status_not_enough_inliers"] == 2
    # a termination_criteria object without an epsilon is refused, one that carries it is TRANSLATED (MULTI.json:627-630 next to :218-223): the loop
    # stops early exactly as with the adapter's own termination_chi_epsilon
    assert (r["threw_on_opaque_termination_criteria"], r["refused_criteria_with_epsilon"], r["threw_on_negative_epsilon"]) == (1, 0, 1)
    assert r["refused_after_reset"] == 0 and 2 <= r["iterations_with_epsilon"] < iters
    assert r["iterations_with_criteria_object"] == r["iterations_with_epsilon"] and r["pose_with_criteria_object"] == r["pose_with_epsilon"]
    oe = po.align(po.aligner_params(iters, termination_chi_epsilon=1e-3), [po.slice_params()], [f], [wl.map_points], x0)
    assert abs(r["iterations_with_epsilon"] - oe["iterations"]) <= 1 and np.abs(np.array(r["pose_with_epsilon"]) - oe["pose"])[:2].max() < POSE_TOL_M
    # enable_inlier_only_runs / keep_only_inlier_correspondences (MULTI.json:606-610) reach the device loop: against the oracle run the same way
    # (start pose through the stand-in's t2v(v2t()) round trip: counts within a few pairs, poses to the tolerance)
    x_off = np.array([x0[0] + 0.15, x0[1] - 0.1, x0[2] + 0.05], np.float32)
    spc = po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=0.05)
    o_plain = po.align(po.aligner_params(iters), [spc], [f], [wl.map_points], x_off, want_pairs=True)
    o_keep = po.align(po.aligner_params(iters, keep_only_inlier_correspondences=True), [spc], [f], [wl.map_points], x_off, want_pairs=True)
    o_runs = po.align(po.aligner_params(iters, keep_only_inlier_correspondences=True, enable_inlier_only_runs=True), [spc], [f], [wl.map_points], x_off, want_pairs=True)
    assert r["plain_iterations"] == iters == o_plain["iterations"] and abs(r["plain_pairs"] - len(o_plain["pairs"][0])) <= 3
    assert abs(r["plain_pairs"] - (r["plain_last_inliers"] + r["plain_last_outliers"])) == 0
    assert r["keep_pairs"] == r["keep_last_inliers"] and abs(r["keep_pairs"] - len(o_keep["pairs"][0])) <= 3 and r["keep_pose"] == r["plain_pose"]
    assert r["inlier_runs_iterations"] == o_runs["iterations"] == 2 * iters and r["inlier_runs_status"] == 0
    assert r["inlier_runs_pairs"] == r["inlier_runs_last_inliers"] and abs(r["inlier_runs_pairs"] - len(o_runs["pairs"][0])) <= 3
    assert np.abs(np.array(r["inlier_runs_pose"]) - o_runs["pose"])[:2].max() < POSE_TOL_M
    # the tracker's three-slice configuration: two laser slices (normal_cos 0.9 + Cauchy 0.01, normal_cos 0.8) and the odometry prior z = x0
    sp0 = po.slice_params(normal_cos=0.9, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.01); sp1 = po.slice_params()
    om = po.align(po.aligner_params(iters, prior_z=x0, prior_omega=np.eye(3, dtype=np.float32)), [sp0, sp1], [f, f], [wl.map_points, wl.map_points], x0)
    dm = np.abs(np.array(r["pose_multi"]) - om["pose"])
    assert r["status_multi"] == 3 and dm[:2].max() < POSE_TOL_M and dm[2] < POSE_TOL_RAD, dm
    no_prior = po.align(po.aligner_params(iters), [sp0, sp1], [f, f], [wl.map_points, wl.map_points], x0)
    assert np.abs(no_prior["pose"] - om["pose"]).max() > 1e-5      # the prior does pull: dropping it (round 1's adapter) would show
    assert r["threw_on_unknown_slice"] == 1
    # mapping siblings
    opr = po.Projector(1081, -math.pi, math.pi, 0.3, 30.0, 0.0)
    robot = synth.invert_poses(x0[None, :].astype(np.float64))[0].astype(np.float32)
    n_clip = len(po.clip_scene(opr, wl.map_points, robot)[0])
    assert abs(r["clipped"] - n_clip) <= 3 and 10 < r["clipped_voxelized"] < r["clipped"] and r["clip_status"] == 1
    n_merge = len(po.merge_scene(opr, wl.map_points, f, robot, 0.2)[0])
    assert abs(r["merged_size"] - n_merge) <= 5 and r["merge_status"] == 1
    # raw-data preprocessor sibling: the reference module's behaviour on unset inputs / foreign topics / null messages, the un-projector it
    # shares with other modules set per message (.cpp:96-101), and the cloud itself bit for bit (class defaults: voxelize 0.02, normals 0.3 / 5)
    want = po.preprocess_scan(po.Preprocessor(721, np.float32(a0), np.float32(a1), 0.3, 20.0, 0.3, 5, 0.02), ranges)
    got = np.fromfile(tmp_path / "prep.bin", np.float32).reshape(-1, 4)
    assert r["prep_status_unset"] == 0 and r["prep_took_other_topic"] == 0 and r["prep_threw_on_null"] == 1 and r["prep_took"] == 1 and r["prep_status"] == 1
    assert r["prep_points"] == len(want) > 300 and np.array_equal(got, want)
    assert abs(r["unprojector_range_max"] - 20.0) < 1e-6 and abs(r["unprojector_range_min"] - 0.3) < 1e-6


def test_cpp_loop_closure_sweep_over_several_contexts(ctx, po, tmp_path):
    """lsm2d_host::LoopClosureSweep / lsm2d_sweep_* (the multi-device candidate loop without Python): 1, 2 and 3 contexts on this one GPU --
    one host thread each, the submap replicated device to device, candidates block-sharded -- must give, bit for bit, the poses,
    information matrices, statuses and last-iteration statistics of ONE lsm2d_align_batch over all candidates; acceptance decisions
    as MULTI.json:979-985."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "sweep_driver")
    lib_dir = os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "host"),
                    os.path.join(ROOT, "tests", "cpp", "sweep_driver.cpp"), "-L" + lib_dir, "-llsm2d_hip", "-Wl,-rpath," + lib_dir, "-pthread", "-o", exe], check=True)
    n_cand, n_unique, iters, tau = 1500, 64, 15, 0.05
    wl = synth.make_workload(n_unique, 50000, seed=21)
    scan_index = (np.arange(n_cand) * 7 % n_unique).astype(np.int32)
    delta = synth.Stream(77, salt=9).uniform(3 * n_cand, -0.05, 0.05).reshape(n_cand, 3)
    x0 = synth.invert_poses(synth.compose_poses(synth.invert_poses(wl.x_true)[scan_index], delta)).astype(np.float32)
    x0[-40:] += np.float32([3.0, -2.0, 0.7])            # hopeless candidates: must be rejected, whatever the device count
    proj = api.PointNormal2fProjectorPolar(1081, -np.float32(math.pi), np.float32(math.pi), 0.3, 30.0)
    al = api.MultiAligner2D(ctx, max_iterations=iters, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(
        api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8), robustifier=api.RobustifierCauchy(tau), min_num_correspondences=10))
    ref = al.compute_batch([api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)], [api.CloudSet(ctx, wl.map_points)], x0, fixed_index=scan_index[None, :], want_stats=True)
    want_acc = ref.loop_closure_accept(500, 0.1, 0.8)
    assert want_acc[:-40].all() and not want_acc[-40:].any()
    last = ref.last_stats()
    for devices in ([0], [0, 0], [0, 0, 0], [0] * 8):      # (eight: the node the driver's scaling run uses -- eight contexts, eight host threads, here on one card)
        d = tmp_path / ("g%d" % len(devices)); d.mkdir()
        wl.scan_points.tofile(d / "scans.bin"); wl.scan_offsets.astype(np.int32).tofile(d / "offsets.bin"); wl.map_points.tofile(d / "map.bin")
        scan_index.tofile(d / "index.bin"); x0.tofile(d / "x0.bin")
        (d / "params.txt").write_text("1081 %d %r 30.0\n" % (iters, tau))
        r = subprocess.run([exe, str(d), str(len(devices))] + [str(v) for v in devices], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert json.loads(r.stdout.strip().splitlines()[-1]) == {"devices": len(devices), "candidates": n_cand}
        pose = np.fromfile(d / "out_pose.bin", np.float32).reshape(n_cand, 3); H = np.fromfile(d / "out_H.bin", np.float32).reshape(n_cand, 3, 3)
        status = np.fromfile(d / "out_status.bin", np.int32); stats = np.fromfile(d / "out_stats.bin", api.STATS_DTYPE)
        acc = np.fromfile(d / "out_accept.bin", np.uint8).astype(bool)
        assert np.array_equal(pose, ref.pose) and np.array_equal(H, ref.information) and np.array_equal(status, ref.status), len(devices)
        assert np.array_equal(stats, last) and np.array_equal(acc, want_acc), len(devices)


def test_sweep_api_error_paths_and_index_defaults(ctx, small_workload):
    """lsm2d_sweep_* through the C ABI directly: a device that does not exist, aligning before the clouds are set, a candidate list
    that needs an index array and has none, an index out of range -- all refused with a message, none of them fatal to the sweep;
    then the two index-free forms (one scan for every candidate, one scan per candidate) against lsm2d_align_batch."""
    import ctypes as C
    from srrg2_laser_slam_2d_amd import _capi
    lib = _capi.load(); wl = small_workload
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    bad = (C.c_int32 * 1)(9999); sw = C.c_void_p()
    assert lib.lsm2d_sweep_create(bad, 1, C.byref(sw)) < 0 and not sw.value
    assert lib.lsm2d_sweep_create(None, 1, C.byref(sw)) == _capi.BAD_ARGUMENT
    assert lib.lsm2d_sweep_create((C.c_int32 * 1)(0), 0, C.byref(sw)) == _capi.BAD_ARGUMENT
    assert lib.lsm2d_sweep_num_devices(None) == 0
    lib.lsm2d_sweep_destroy(None)
    devs = (C.c_int32 * 2)(0, 0)
    assert lib.lsm2d_sweep_create(devs, 2, C.byref(sw)) == 0 and lib.lsm2d_sweep_num_devices(sw) == 2
    try:
        n = len(wl.x0); its = 6
        ap = _capi.AlignerParams(its, 10, 0.0)
        sp = api.make_slice_params(projector=_projector(), robustifier=0, min_num_correspondences=10)
        x0 = np.ascontiguousarray(wl.x0, np.float32)
        pose = np.zeros((n, 3), np.float32); status = np.full(n, -7, np.int32); iters = np.zeros(n, np.int32)
        call = lambda k, idx, x: lib.lsm2d_sweep_align(sw, C.byref(ap), C.byref(sp), k, idx, P(x), P(pose), None, P(status), P(iters), None)
        # nothing set yet
        assert call(n, None, x0) == _capi.BAD_ARGUMENT and b"set_map" in lib.lsm2d_sweep_last_error(sw)
        scans = np.ascontiguousarray(wl.scan_points); offs = np.ascontiguousarray(wl.scan_offsets, np.int32); mp = np.ascontiguousarray(wl.map_points)
        assert lib.lsm2d_sweep_set_scans(sw, P(scans), P(offs), n) == 0
        assert call(n, None, x0) == _capi.BAD_ARGUMENT                          # still no map
        assert lib.lsm2d_sweep_set_map(sw, None, 10) == _capi.BAD_ARGUMENT
        assert lib.lsm2d_sweep_set_scans(sw, P(scans), None, n) == _capi.BAD_ARGUMENT
        assert call(n, None, x0) == _capi.BAD_ARGUMENT                          # the refused calls changed nothing: still no map
        assert lib.lsm2d_sweep_set_scans(sw, P(scans), P(offs), n) == 0 and lib.lsm2d_sweep_set_map(sw, P(mp), len(mp)) == 0
        # n scans, fewer candidates, no index array
        assert call(n - 1, None, x0) == _capi.BAD_ARGUMENT and b"scan_index" in lib.lsm2d_sweep_last_error(sw)
        # index out of range on the SECOND device's shard only: the whole call fails and says which device
        idx = np.arange(n, dtype=np.int32); idx[-1] = n
        assert call(n, P(idx), x0) == _capi.BAD_ARGUMENT and b"device 1" in lib.lsm2d_sweep_last_error(sw)
        assert call(0, None, x0) == 0                                           # empty sweep
        # one scan per candidate, no index array: candidate i of the second shard uses scan lo + i, not scan i
        al = _aligner(ctx, its=its)
        want = al.compute_batch([api.CloudSet(ctx, scans, offs)], [api.CloudSet(ctx, mp)], x0)
        assert call(n, None, x0) == 0
        assert np.array_equal(pose, want.pose) and np.array_equal(status, want.status) and np.array_equal(iters, want.iterations)
        # one scan for every candidate
        one = np.ascontiguousarray(scans[offs[2]:offs[3]]); o1 = np.array([0, len(one)], np.int32)
        xs = np.ascontiguousarray(np.repeat(x0[2:3], 5, axis=0) + np.linspace(0, 0.02, 5, dtype=np.float32)[:, None])
        assert lib.lsm2d_sweep_set_scans(sw, P(one), P(o1), 1) == 0
        assert call(5, None, xs) == 0
        want = al.compute_batch([api.CloudSet(ctx, one)], [api.CloudSet(ctx, mp)], xs)
        assert np.array_equal(pose[:5], want.pose) and np.array_equal(status[:5], want.status)
    finally:
        lib.lsm2d_sweep_destroy(sw)


def test_cpp_stream_step_through_the_bare_c_abi_and_the_mirror_class(ctx):
    """tests/cpp/stream_step_bench.cpp: the streamed pipeline (lsm2d_preprocess_scans_refill -> lsm2d_align_batch_begin -> lsm2d_align_batch_wait one step behind,
    ranges in pinned host memory) driven from C++ -- the reference's host language -- through the bare C ABI, and again through the C++ mirror's
    LaserMessageBatchStream (host/lsm2d.hpp) on a context of its own: every batch that comes out is BITWISE the synchronous calls' on the same ranges."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("stream_step_bench", os.path.join(ROOT, "tests", "bench", "stream_step_bench.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    for ahead in (1, 0):      # three scan sets, the next step's scans refilled behind this step's begin (what include/lsm2d.h recommends); two sets, refill just before begin
        r = mod.run(steps=14, warmup=5, scans=96, map_points=20000, iterations=10, beams=721, batches=3, seed=5, ahead=ahead)
        assert r["refill_ahead"] == ahead and r["steps_checked_bitwise"] == 14 + 5 - 1 and r["steps_that_differed"] == 0
        assert r["mirror_batches_checked"] == 2 * 3 + 1 and r["mirror_batches_that_differed"] == 0
        assert r["status_ok_batch0"] >= 90 and r["alignments"] == 96


def test_cpp_host_mirror_driver(ctx, po, small_workload, tmp_path):
    """The header-only C++ mirror (srrg2_laser_slam_2d_amd/host/lsm2d.hpp), built with plain g++ and driven like
    apps/visual_test_correspondence_finder_projective_2d.cpp / apps/visual_test_aligner_2d.cpp."""
    import os
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "host_mirror_driver")
    lib_dir = os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "host"),
                    os.path.join(ROOT, "tests", "cpp", "host_mirror_driver.cpp"), "-L" + lib_dir, "-llsm2d_hip", "-Wl,-rpath," + lib_dir, "-o", exe],
                   check=True)
    wl = small_workload
    f = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]
    f.tofile(tmp_path / "fixed.bin"); wl.map_points.tofile(tmp_path / "moving.bin")
    x0 = wl.x0[0]
    out = subprocess.run([exe, str(tmp_path / "fixed.bin"), str(tmp_path / "moving.bin"), repr(float(x0[0])), repr(float(x0[1])), repr(float(x0[2])), "1081", "20"],
                         check=True, capture_output=True, text=True, timeout=120).stdout
    r = json.loads(out)
    want = po.find(po.slice_params(), f, wl.map_points, x0)
    assert r["threw_on_missing_inputs"] == 1
    assert np.array_equal(np.array(r["pairs"], np.int32).reshape(-1, 2), want)
    o = po.align(po.aligner_params(20), [po.slice_params()], [f], [wl.map_points], x0)
    d = np.abs(np.array(r["pose"]) - o["pose"])
    assert r["status"] == 0 and r["iterations"] == 20 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    # round 4 through the C++ mirror: stored correspondences (their host-side digest equals the last iteration's), kept inliers, the second loop
    spc = po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=2e-5)
    o_all = po.align(po.aligner_params(20, device_order=True), [spc], [f], [wl.map_points], x0, want_pairs=True)
    o_run = po.align(po.aligner_params(20, device_order=True, enable_inlier_only_runs=True, keep_only_inlier_correspondences=True), [spc], [f], [wl.map_points], x0, want_pairs=True)
    assert r["digest_matches"] == 1 and r["n_all"] == len(o_all["pairs"][0]) and r["iterations_with_inlier_runs"] == o_run["iterations"] == 40
    assert r["n_kept"] == len(o_run["pairs"][0]) == r["last_inliers_with_inlier_runs"] == o_run["stats"][-1].n_in
    # the other finders and the mapping classes of the C++ mirror give the oracle's counts on the same inputs
    assert r["n_nn"] == len(po.find(po.slice_params(finder=po.FINDER_NN, max_distance=0.3), f, wl.map_points, x0))
    assert r["n_kdtree"] == len(po.find(po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.3, kd_max_leaf_range=0.02, kd_min_leaf_points=9), f, wl.map_points, x0))
    assert r["n_distmap"] == len(po.find(po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.1), f, wl.map_points, x0))
    xi = synth.invert_poses(x0[None, :].astype(np.float64))[0]
    c, s_ = math.cos(float(x0[2])), math.sin(float(x0[2]))
    sensor_in_map = np.float32([-(np.float32(c) * x0[0] + np.float32(s_) * x0[1]), -(-np.float32(s_) * x0[0] + np.float32(c) * x0[1]), -x0[2]])
    opr = po.Projector(1081, -math.pi, math.pi, 0.3, 30.0, 0.0)
    oclip, _ = po.clip_scene(opr, wl.map_points, sensor_in_map)
    omerge, ocounts = po.merge_scene(opr, wl.map_points, f, sensor_in_map, 0.2)
    assert abs(r["n_clipped"] - len(oclip)) <= 2 and abs(r["merged_size"] - len(omerge)) <= 2     # host-side inverse differs in the last bit


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


def _nn_aligner(ctx, md=0.5, its=20):
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=md), min_num_correspondences=10))
    return al


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


def test_loop_closure_sweep_acceptance(ctx, po):
    """Row f3: a sweep of candidate (scan, initial guess) pairs against one submap, as MultiLoopDetectorBruteForce2D does with
    relocalize_aligner (30 iterations, Cauchy 0.05, point_distance 1.414: MULTI.json:572-630,771-784), then the acceptance
    test of MULTI.json:979-985.  Good guesses must be accepted, hopeless ones rejected; decisions equal the oracle's."""
    wl = synth.make_workload(12, 60000, seed=12)
    x0 = wl.x0.copy()
    x0[8:] += np.float32([3.0, -2.0, 0.7])                      # candidates 8..11: wrong place
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    al = api.MultiAligner2D(ctx, max_iterations=30, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(
        api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=1.414, normal_cos=0.8), robustifier=api.RobustifierCauchy(0.05),
        min_num_correspondences=10))
    res = al.compute_batch([api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)], [api.CloudSet(ctx, wl.map_points)], x0, want_stats=True)
    acc = res.loop_closure_accept(300, 0.1, 0.8)
    osp = po.slice_params(canvas_cols=721, range_max=20.0, point_distance=1.414, robustifier=po.ROBUST_CAUCHY, chi_threshold=0.05)
    xo, _, status, last = po.align_batch(po.aligner_params(30), osp, wl.scan_points, wl.scan_offsets, wl.map_points, x0)
    want = np.array([status[i] == 0 and last[i].n_in >= 300 and last[i].chi_in / max(last[i].n_in, 1) <= 0.1 and
                     last[i].n_in / max(last[i].n_corr, 1) >= 0.8 for i in range(12)])
    assert np.array_equal(acc, want)
    assert acc[:8].all() and not acc[8:].any()


def test_configs3_full_size_loop_closure_sweep(ctx, po):
    """BASELINE configs[3] at its size on one GPU: 65 536 candidate (scan, initial guess) pairs -- 2 048 distinct scans chosen
    through the index array, as MultiLoopDetectorBruteForce2D's candidate loop would (MULTI.json:964-986) -- against one 100k-point
    submap, Cauchy tau 0.05 (MULTI.json:957-962).  Size-independent properties over the whole sweep (generating pose,
    equivariance under a permutation of the candidates, run-to-run bits, acceptance decisions) and 16 sampled candidates against
    the oracle: within the north_star tolerance of the reference-order mirror, bit-identical to the device-order mirror."""
    n_cand, n_unique, iters = 65536, 2048, 20
    world = synth.make_world(3)
    wl = synth.make_workload(n_unique, 100000, seed=3, world=world)
    scan_index = (np.arange(n_cand) % n_unique).astype(np.int32)
    st = synth.Stream(4242, salt=9)
    delta = st.uniform(3 * n_cand, -0.05, 0.05).reshape(n_cand, 3)
    x_true = wl.x_true[scan_index]
    x0 = synth.invert_poses(synth.compose_poses(synth.invert_poses(wl.x_true)[scan_index], delta)).astype(np.float32)
    proj = api.PointNormal2fProjectorPolar(1081, -math.pi, math.pi, 0.3, 30.0)
    al = api.MultiAligner2D(ctx, max_iterations=iters, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(
        api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8), robustifier=api.RobustifierCauchy(0.05),
        min_num_correspondences=10))
    scans = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); submap = api.CloudSet(ctx, wl.map_points)
    res = al.compute_batch([scans], [submap], x0, fixed_index=scan_index[None, :], want_stats=True)
    assert ctx.get_option("last_align_path") == 1                      # the throughput kernel
    # (1) noise-free data: every candidate converges to the pose its scan was rendered from
    err = np.abs(res.pose - x_true); err[:, 2] = np.abs((err[:, 2] + np.pi) % (2 * np.pi) - np.pi)
    assert (res.status == 0).all() and err[:, :2].max() < 1e-4 and err[:, 2].max() < 1e-4, (err[:, :2].max(), err[:, 2].max())
    assert (res.iterations == iters).all()
    # (2) the acceptance test of the sweep's consumer (MULTI.json:979-985): all of these are true closures
    assert res.loop_closure_accept(500, 0.1, 0.8).all()
    # (3) run-to-run: the same bits
    res2 = al.compute_batch([scans], [submap], x0, fixed_index=scan_index[None, :])
    assert np.array_equal(res.pose, res2.pose) and np.array_equal(res.information, res2.information)
    # (4) a permutation of the candidates permutes the results, bit for bit (an alignment does not depend on its neighbours)
    perm = np.random.default_rng(5).permutation(n_cand)
    resp = al.compute_batch([scans], [submap], x0[perm], fixed_index=scan_index[perm][None, :])
    assert np.array_equal(resp.pose, res.pose[perm]) and np.array_equal(resp.information, res.information[perm])
    # (5) 16 sampled candidates against the oracle
    osp = po.slice_params(robustifier=po.ROBUST_CAUCHY, chi_threshold=0.05)
    for i in np.random.default_rng(6).choice(n_cand, 16, replace=False):
        sc = wl.scan_points[wl.scan_offsets[scan_index[i]]:wl.scan_offsets[scan_index[i] + 1]]
        ref = po.align(po.aligner_params(iters), [osp], [sc], [wl.map_points], x0[i])
        d = np.abs(res.pose[i] - ref["pose"])
        assert ref["status"] == 0 and d[:2].max() < 1e-4 and d[2] < 1e-4, (i, d)
        dev = po.align(po.aligner_params(iters, device_order=True), [osp], [sc], [wl.map_points], x0[i])
        assert np.array_equal(res.pose[i], dev["pose"]) and np.array_equal(res.information[i], dev["H"]), i


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


def test_bench_strong_scaling_leg_runs_over_rccl_on_one_gpu(tmp_path):
    """bench.py's N > 1 leg for configs[3] (shard the candidates, RCCL broadcast of the submap, all_gather of the poses, the
    cross-rank bit check) executed on hardware with a world of one rank: LSM2D_BENCH_FORCE_DIST=1 initialises the nccl (= RCCL)
    process group and takes every collective the 8-GPU run takes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSM2D_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--total-candidates", "65536", "--unique-scans", "2048",
                        "--cauchy", "0.05", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["parity_ok"] and d["ranks_seen"] == 1 and d["scaling"] == "strong" and d["config"]["alignments_per_gpu"] == 65536
    assert d["cross_rank_check"].startswith("1 of 1 ranks"), d["cross_rank_check"]
    assert d["value"] > 10000 and d["max_pose_err_m"] < 1e-4


def test_bench_four_ranks_share_the_gpu_weak_and_strong(tmp_path):
    """bench.py launched as the round-end driver launches it for N > 1 (torch.distributed.run, one process per rank) with FOUR ranks
    on THIS one GPU (the pool allows six processes on a card: this test process, the launcher and four ranks -- five ranks were killed by its process guard;
    round 4 rehearsed three): RCCL refuses two
    ranks on a device, so the transport is gloo (LSM2D_BENCH_BACKEND) -- everything else is the N-GPU
    run: per-rank scans, the submap broadcast from rank 0, sharding, barrier-bracketed timing with the maximum over ranks, the cross-rank
    bit check, every rank pinned to its own cores and keeping its own note file.  Weak scaling (the default line) and the strong-scaling sweep
    of configs[3] at a reduced size."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LSM2D_BENCH_BACKEND="gloo", LSM2D_BENCH_RANK_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    NR = 4
    for extra, scaling, per_rank in ((["--scans", "300"], "weak", 300), (["--total-candidates", "3001", "--unique-scans", "256", "--cauchy", "0.05"], "strong", None)):
        # weak: through the launcher, as the driver does; strong: the PLAIN command -- bench.py finds no WORLD_SIZE and starts its three ranks itself
        # (round 3's plain `--gpus N` silently ran one rank and printed n_gpus: 1)
        launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(NR), "--master-addr", "127.0.0.1", "--master-port", "29541"] if per_rank else [sys.executable]
        r = subprocess.run(launcher + [os.path.join(root, "bench.py"), "--gpus", str(NR), "--steps", "3", "--warmup", "1", "--spinup-s", "0.05", "--no-cpu-baseline"] + extra,
                           env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints, the others stay silent
        d = json.loads(lines[0])
        assert d["n_gpus"] == NR and d["ranks_seen"] == NR and d["scaling"] == scaling and d["parity_ok"], d
        assert d["cross_rank_check"].startswith("%d of %d ranks" % (NR, NR)), d["cross_rank_check"]
        assert len(d["ms_per_step_per_rank"]) == NR and max(d["ms_per_step_per_rank"]) <= d["ms_per_step"] * 1.001      # the line's time is the slowest rank's
        # every rank's own note file reached "done" with a parity verdict; the affinity masks are disjoint (when the box has the cores) and cover what rank 0 may use
        notes = [json.load(open(tmp_path / ("bench_rank%d.json" % k))) for k in range(NR)]
        assert all(nt["stage"] == "done" and nt["parity_ok"] and nt["world"] == NR for nt in notes)
        cores = [c for nt in notes for c in nt["cpu_affinity"]]
        assert len(set(cores)) == len(cores) or len(notes[0]["cpu_affinity"]) < NR
        assert ("strong_scaling_gather" in d) == (scaling == "strong")                                                   # ... and the sweep's gather is inside it
        if per_rank:
            assert d["config"]["alignments_per_gpu"] == per_rank and abs(d["value"] * d["ms_per_step"] * 1e-3 - NR * per_rank) < 1e-6 * NR * per_rank
        else:
            assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 3001) < 1e-2       # the whole sweep per step, whatever the shard sizes
            sh = d["sharding"]                                                    # sharded by estimated work: balanced to within a candidate's worth, never worse than by count
            assert sh["by"] == "work" and sum(sh["candidates_per_rank"]) == 3001 and sh["work_max_over_mean"] <= min(1.01, sh["work_max_over_mean_if_sharded_by_count"] + 1e-9)
            assert d["ms_per_step_rank_max"] == max(d["ms_per_step_per_rank"])
        assert d["max_pose_err_m"] < 1e-4


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


def test_abi_error_paths_and_limits(ctx, small_workload):
    """Call-level errors come back as negative codes (never exceptions / crashes across the ABI); limits are enforced."""
    import ctypes as C
    from srrg2_laser_slam_2d_amd import _capi
    lib = ctx._lib
    wl = small_workload
    m = api.CloudSet(ctx, wl.map_points); s = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    # projector validation
    for bad in (api.PointNormal2fProjectorPolar(0), api.PointNormal2fProjectorPolar(721, 1.0, -1.0), api.PointNormal2fProjectorPolar(721, -3.14, 3.14, 5.0, 1.0)):
        with pytest.raises(api.Lsm2dError) as ei:
            bad.compute(ctx, m)
        assert ei.value.code == _capi.BAD_ARGUMENT
    # a canvas that cannot fit the 160 KiB LDS of a CU
    with pytest.raises(api.Lsm2dError) as ei:
        api.PointNormal2fProjectorPolar(40000).compute(ctx, m)
    assert ei.value.code == _capi.CAPACITY_EXCEEDED
    # the largest canvas that does fit still works (and matches a smaller run on the columns they share a boundary with)
    src, depth, _ = api.PointNormal2fProjectorPolar(16000, -math.pi, math.pi, 0.3, 30.0).compute(ctx, m, wl.x0[0])
    assert (src >= 0).sum() > 1000
    # cloud index out of range, unknown finder, cloud sets of another size than the batch
    f = api.CorrespondenceFinderProjective2f(ctx, _projector())
    f.setFixed(s, 99); f.setMoving(m); f.setLocalMapInSensor([0, 0, 0])
    with pytest.raises(api.Lsm2dError):
        f.compute()
    al = _aligner(ctx)
    with pytest.raises(api.Lsm2dError):
        al.compute_batch([s], [m], wl.x0[:3])                  # 6 clouds, 3 alignments, no index array
    with pytest.raises(api.Lsm2dError):
        al.compute_batch([s], [m], wl.x0, fixed_index=np.full((1, len(wl.x0)), 77, np.int32))
    sp = api.make_slice_params(finder=7)
    n = C.c_int32(0); out = np.zeros((10, 2), np.int32)
    rc = lib.lsm2d_find_correspondences(ctx.handle, C.byref(sp), s.handle, 0, m.handle, 0, np.zeros(3, np.float32).ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p), 10, C.byref(n))
    assert rc == _capi.BAD_ARGUMENT and b"finder" in lib.lsm2d_last_error(ctx.handle)
    # output capacity too small: the count is still reported
    sp = api.make_slice_params(projector=_projector())
    rc = lib.lsm2d_find_correspondences(ctx.handle, C.byref(sp), s.handle, 0, m.handle, 0, wl.x0[0].ctypes.data_as(C.c_void_p),
                                        out.ctypes.data_as(C.c_void_p), 10, C.byref(n))
    assert rc == _capi.CAPACITY_EXCEEDED and n.value > 10
    # null handles
    assert lib.lsm2d_synchronize(None) == _capi.BAD_ARGUMENT
    assert lib.lsm2d_cloudset_num_points(None) == 0
    # empty batch is a no-op
    r = al.compute_batch([s], [m], np.zeros((0, 3), np.float32))
    assert len(r.pose) == 0
    # more than 4 slices is rejected
    al5 = api.MultiAligner2D(ctx)
    for _ in range(5):
        al5.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector())))
    with pytest.raises(api.Lsm2dError):
        al5.compute_batch([m] * 5, [m] * 5, np.zeros((1, 3), np.float32))
    # Cauchy with a non-positive threshold
    alc = _aligner(ctx, robustifier=api.RobustifierCauchy(0.0))
    with pytest.raises(api.Lsm2dError):
        alc.compute_batch([s], [m], wl.x0)
    # kernel timing is opt-in at the ABI: a context without it gives the same results and refuses lsm2d_last_kernel_ms
    quiet = api.Context(0, kernel_timing=False)
    try:
        alq = _aligner(quiet)
        rq = alq.compute_batch([api.CloudSet(quiet, wl.scan_points, wl.scan_offsets)], [api.CloudSet(quiet, wl.map_points)], wl.x0)
        rt_ = al.compute_batch([s], [m], wl.x0)
        assert np.array_equal(rq.pose, rt_.pose) and rq.kernel_ms == 0.0 and rt_.kernel_ms > 0.0
        with pytest.raises(api.Lsm2dError) as ei:
            quiet.last_kernel_ms()
        assert ei.value.code == _capi.BAD_ARGUMENT
        quiet.set_option("kernel_timing", 1)
        alq.compute_batch([api.CloudSet(quiet, wl.scan_points, wl.scan_offsets)], [api.CloudSet(quiet, wl.map_points)], wl.x0)
        assert quiet.last_kernel_ms() > 0.0
    finally:
        quiet.close()


def test_no_device_memory_is_left_behind(small_workload):
    """Contexts, cloud sets, the finders' cached structures (grids, distance maps, lane-chunked copies), reserved sets that grow, sweeps:
    created, used and destroyed 25 times over -- the device's free memory ends where it started (64 MB of slack for the runtime's own pools)."""
    import ctypes as C
    import torch
    from srrg2_laser_slam_2d_amd import _capi
    lib = _capi.load(); wl = small_workload
    scan = wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]

    def cycle():
        c = api.Context(0)
        scans = api.CloudSet(c, wl.scan_points, wl.scan_offsets); mp = api.CloudSet(c, wl.map_points)
        for f in (api.CorrespondenceFinderKDTree2D(c, max_distance_m=0.3), api.CorrespondenceFinderNN2D(c, max_distance_m=0.5, resolution=0.05),
                  api.CorrespondenceFinderProjective2f(c, _projector())):
            al = api.MultiAligner2D(c, max_iterations=3, min_num_inliers=10)
            al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(f, min_num_correspondences=10))
            al.compute_batch([scans], [mp], wl.x0)                                           # structures over the scans
            al.compute_batch([mp], [scans], synth.invert_poses(wl.x0.astype(np.float64)).astype(np.float32)) if not isinstance(f, api.CorrespondenceFinderProjective2f) else None
        grow = api.CloudSet.reserved(c, 40000); grow.upload(wl.map_points)
        m = api.MergerProjective2D(c, _projector(), 0.2); m.setScene(grow); m.setMeasurement(scan); m.setMeasurementInScene(synth.invert_poses(wl.x_true[:1])[0].astype(np.float32)); m.compute()
        clip = api.SceneClipperProjective2D(c, _projector(), voxelize_resolution=0.0); clip.setFullScene(grow); clip.setRobotInLocalMap(synth.invert_poses(wl.x_true[:1])[0].astype(np.float32)); clip.compute()
        sw = C.c_void_p(); assert lib.lsm2d_sweep_create((C.c_int32 * 2)(0, 0), 2, C.byref(sw)) == 0
        pts = np.ascontiguousarray(wl.map_points)
        assert lib.lsm2d_sweep_set_map(sw, pts.ctypes.data_as(C.c_void_p), len(pts)) == 0
        lib.lsm2d_sweep_destroy(sw)
        c.close()

    cycle(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free0 - free1 < 64 << 20, (free0, free1)


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


def test_sets_may_outlive_their_context(ctx, small_workload):
    """lsm2d_destroy orphans the sets still alive on it: destroying them afterwards is fine, using them is an error, and nothing
    of it disturbs another context."""
    from srrg2_laser_slam_2d_amd import _capi
    import ctypes as C
    lib = _capi.load(); wl = small_workload
    c2 = C.c_void_p(); assert lib.lsm2d_create(0, None, C.byref(c2)) == 0
    pts = np.ascontiguousarray(wl.map_points[:1000]); h = C.c_void_p(); r = C.c_void_p()
    assert lib.lsm2d_cloudset_create(c2, pts.ctypes.data_as(C.c_void_p), None, 1, len(pts), C.byref(h)) == 0
    assert lib.lsm2d_cloudset_create_reserved(c2, 2048, C.byref(r)) == 0
    assert lib.lsm2d_cloudset_upload(r, pts.ctypes.data_as(C.c_void_p), 500) == 0        # left pending on purpose
    lib.lsm2d_destroy(c2)
    out = np.empty((1000, 4), np.float32); n = C.c_int64(0)
    assert lib.lsm2d_cloudset_download(r, 0, out.ctypes.data_as(C.c_void_p), 1000, C.byref(n)) < 0
    assert lib.lsm2d_cloudset_upload(r, pts.ctypes.data_as(C.c_void_p), 10) < 0
    assert lib.lsm2d_cloudset_num_points(h) == 1000                                      # host-side knowledge survives
    lib.lsm2d_cloudset_destroy(h); lib.lsm2d_cloudset_destroy(r)
    al = _aligner(ctx, 361, its=5)                                                        # the session's context is untouched
    res = al.compute_batch([wl.scan_points[wl.scan_offsets[0]:wl.scan_offsets[1]]], [wl.map_points], wl.x0[:1])
    assert res.status[0] == 0


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


def _ranges_in_pose_out_step(ctx, po):
    world = synth.make_world(6)
    a0, a1 = -2.34747, 2.35619
    S = [np.float32([0.2, 0.1, 0.1]), np.float32([-0.3, 0.0, math.pi])]
    robot = synth.sample_poses(world, 1, seed=12)[0]
    sensors = [synth.compose_poses(robot[None, :], s[None, :].astype(np.float64)) for s in S]
    ranges = [synth.make_scan_ranges(world, sp, n_beams=721, angle_min=a0, angle_max=a1, noise_sigma=0.004, seed=40 + i)[0] for i, sp in enumerate(sensors)]
    proj = api.PointNormal2fProjectorPolar(721, -math.pi, math.pi, 0.3, 20.0)
    opr = po.Projector(721, -math.pi, math.pi, 0.3, 20.0, 0.0)
    m = synth.make_map(world, 20000, noise_sigma=0.004, seed=2)
    guess = synth.compose_poses(robot[None, :], np.array([[0.03, -0.02, 0.02]]))[0].astype(np.float32)
    # --- device
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=20.0, voxelize_resolution=0.02)
    sets = [api.CloudSet.reserved(ctx, 1024), api.CloudSet.reserved(ctx, 1024)]
    for i in range(2):
        pre.setRawData(ranges[i], a0, a1, 0.0, 30.0); pre.compute_into(sets[i])
    local_map = api.CloudSet.reserved(ctx, 30000); local_map.upload(m)
    clipper = api.SceneClipperProjective2D(ctx, proj, asynchronous=True, voxelize_resolution=0.0); clipper.setFullScene(local_map)
    clipper.setRobotInLocalMap(guess); clipper.setSensorInRobot(S[0])
    clipped = clipper.compute()
    al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
    for i, s in enumerate(S):
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2DWithSensor(
            api.CorrespondenceFinderProjective2f(ctx, proj, 0.5, 0.8), sensor_in_robot=s, min_num_correspondences=5,
            fixed_slice_name="points_%d" % i, moving_slice_name="points"))
    al.setFixed({"points_0": sets[0], "points_1": sets[1]}); al.setMoving({"points": clipped}); al.setMovingInFixed([0, 0, 0])
    assert al.compute() == 0
    est = synth.compose_poses(guess[None, :].astype(np.float64), synth.invert_poses(al.movingInFixed()[None, :].astype(np.float64)))[0]
    merger = api.MergerProjective2D(ctx, proj, 0.2, asynchronous=True); merger.setScene(local_map)
    for i, s in enumerate(S):
        merger.setMeasurement(sets[i]); merger.setMeasurementInScene(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0]); merger.compute()
    # --- oracle
    pp = po.Preprocessor(721, a0, a1, 0.3, 20.0, 0.3, 5, 0.02)
    meas = [po.preprocess_scan(pp, r) for r in ranges]
    for i in range(2):
        assert np.array_equal(sets[i].download(), meas[i]) and len(meas[i]) > 200
    oclip, _ = po.clip_scene(opr, m, guess, S[0])
    osl = [po.slice_params(canvas_cols=721, range_max=20.0, normal_cos=0.8, min_num_correspondences=5, sensor_in_robot=tuple(s)) for s in S]
    r = po.align(po.aligner_params(10), osl, meas, [oclip, oclip], np.zeros(3, np.float32))
    d = np.abs(al.movingInFixed() - r["pose"])
    assert r["status"] == 0 and d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD
    assert np.abs(est - robot)[:2].max() < 0.03
    host_map = m
    for i, s in enumerate(S):       # merged at the DEVICE's estimate so that the maps can be compared bit for bit
        host_map, _ = po.merge_scene(opr, host_map, meas[i], np.float32(synth.compose_poses(est[None, :], s[None, :].astype(np.float64))[0]), 0.2)
    assert local_map.n_points == len(host_map) and np.array_equal(local_map.download(), host_map)


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


def test_randomised_parameters_finder_and_aligner(ctx, po):
    """Fuzz the bit-exact contract over the parameter space the ABI accepts: asymmetric fields of view, odd canvas sizes,
    column rounding, tight and wide gates, all three finders, Cauchy on/off, sensor extrinsics -- finder pairs must equal
    the oracle's exactly, aligner poses within the north_star tolerance whenever the oracle succeeds."""
    import os
    n_trials = int(os.environ.get("LSM2D_FUZZ_TRIALS", "36")); seed = int(os.environ.get("LSM2D_FUZZ_SEED", "2024"))      # soak: more trials, other seeds
    only = int(os.environ.get("LSM2D_FUZZ_ONLY", "-1"))          # reproduce one trial of a soak run, verbosely
    checked_pairs = checked_poses = soft = sets_differ = seq_bitwise = 0
    env = _Envelope("parameters", seed)
    for spec in fuzz_cases.parameter_trials(seed, n_trials):      # the draws: tests/fuzz_cases.py (shared with tests/replay_violators.py)
        trial, finder, m, scan, x0, n_map, beams = spec["trial"], spec["finder"], spec["map"], spec["scan"], spec["x0"], spec["n_map"], spec["beams"]
        a0, a1, cols, off, rmin, rmax, pd, nc, md, res = (spec[k] for k in ("a0", "a1", "cols", "off", "rmin", "rmax", "pd", "nc", "md", "res"))
        cauchy, tau, mc, S, its, min_inl = (spec[k] for k in ("cauchy", "tau", "mc", "S", "its", "min_inl"))
        if only >= 0 and trial != only:
            continue
        f, osp = fuzz_cases.parameter_finder(ctx, spec)
        f.setFixed(scan); f.setMoving(m); f.setLocalMapInSensor(x0)
        got = f.compute(); want = po.find(osp, scan, m, x0)
        assert np.array_equal(got, want), (trial, finder, len(got), len(want))
        checked_pairs += len(want)
        # aligner with the same finder
        fuzz_cases.parameter_aligner_slice(po, spec, osp)
        al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=min_inl)
        al.param_slice_processors.append(fuzz_cases.parameter_slice_processor(spec, f))
        res_g = al.compute_batch([scan], [m], x0[None, :], want_stats=True)
        r = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers), [osp], [scan], [m], x0)
        rd = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers), [osp], [scan], [m], x0.astype(np.float64), double=True)
        rt = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers, device_order=True), [osp], [scan], [m], x0)
        if only < 0:
            _assert_bitwise_equal_to_device_order_oracle(res_g, 0, rt, ("trial=%d" % trial, finder))       # EVERY trial, well-posed or not
            if finder == 0:      # the batch kernel with its exact culling against the fixed canvas (the call above ran the latency kernel, which has none):
                ctx.set_option("align_path", 1)      # random fields of view, column rounding and gates through chunk_may_matter
                try:
                    res_c = al.compute_batch([scan], [m], x0[None, :], want_stats=True)
                finally:
                    ctx.set_option("align_path", 0)
                _assert_bitwise_equal_to_device_order_oracle(res_c, 0, rt, ("trial=%d culled" % trial, finder))
            # Round 6: with "sum_order" 1 the device adds pair after pair, the reference's order -- and equals the SEQUENTIAL fp32 oracle `r` (the restatement written
            # from the reference's files, not after the device) bit for bit, in EVERY trial, well-posed or not: status, iterations, pose, information matrix, statistics, digests
            ctx.set_option("sum_order", 1)
            try:
                res_s = al.compute_batch([scan], [m], x0[None, :], want_stats=True)
            finally:
                ctx.set_option("sum_order", 0)
            _assert_bitwise_equal_to_device_order_oracle(res_s, 0, r, ("trial=%d sum_order 1" % trial, finder))
            seq_bitwise += 1
        if only >= 0:
            print("trial", trial, dict(finder=finder, n_map=n_map, beams=beams, cols=cols, off=off, a0=a0, a1=a1, rmin=rmin, rmax=rmax, pd=pd, nc=nc, md=md, res=res,
                                       cauchy=cauchy, tau=tau, mc=mc, S=S, its=its, min_inl=min_inl, x0=x0.tolist()))
            print(" gpu  status", res_g.status[0], "its", res_g.iterations[0], "n_corr", res_g.stats[0]["n_correspondences"][:its].tolist(), "pose", res_g.pose[0].tolist())
            print(" f32  status", r["status"], "its", r["iterations"], "n_corr", [st.n_corr for st in r["stats"]], "pose", r["pose"].tolist())
            print(" f64  status", rd["status"], "its", rd["iterations"], "n_corr", [st.n_corr for st in rd["stats"]], "pose", rd["pose"].tolist())
            print(" H gpu", res_g.H[0].ravel().tolist()); print(" H f32", r["H"].ravel().tolist()); print(" H f64", rd["H"].ravel().tolist())
        assert res_g.stats[0]["n_correspondences"][0] == r["stats"][0].n_corr, ("trial=%d" % trial, finder, S)      # first iteration: same pose, same pairs -- always
        # the STRICT class: the two oracles agree on status and iteration count, sit within 2.5e-5 of each other, and the device used the sequential fp32 oracle's
        # pairs in every iteration (digests) -- north_star's bar against that oracle.  Everything else (degenerate geometry, runaway iterations, pair sets that
        # part ways) is held to the ENVELOPE of the reference's own arithmetic around the fp64 truth (round 5: no flat centimetre, nothing skipped)
        dd = np.abs(r["pose"].astype(np.float64) - rd["pose"]); dd[2] = abs((dd[2] + math.pi) % (2 * math.pi) - math.pi)
        agree = r["status"] == rd["status"] == 0 and r["iterations"] == rd["iterations"] and res_g.status[0] == 0 and res_g.iterations[0] == r["iterations"]
        same_sets = bool(agree) and _same_correspondence_sets(res_g.stats[0], r["stats"], r["iterations"])      # exact: the iterations' pair digests
        if same_sets and 4.0 * dd[:2].max() <= POSE_TOL_M and 4.0 * dd[2] <= POSE_TOL_RAD:
            d = np.abs(res_g.pose[0] - r["pose"]); d[2] = abs((d[2] + math.pi) % (2 * math.pi) - math.pi)
            assert d[:2].max() < POSE_TOL_M and d[2] < POSE_TOL_RAD, (trial, finder, d, dd)
            checked_poses += 1
            continue
        if r["status"] == rd["status"] and r["status"] != 0:      # both oracles fail alike: the device's status is its mirror's (bitwise above)
            continue
        rr = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers), [osp], [scan], [m], x0, double="ref")
        env.check((trial, 0, "finder %d" % finder), res_g.pose[0], int(res_g.status[0]), r, rd, rr,
                  perturbed=lambda: [po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers), [osp], [scan], [m], xp) for xp in _Envelope.one_ulp_starts(x0)])
        sets_differ += int(bool(agree) and not same_sets)
        soft += 1
        checked_poses += 1
    if only >= 0:
        return
    print("fuzz, sum_order 1: %d of %d aligner runs BITWISE equal to the sequential fp32 oracle (status, iterations, pose, information matrix, statistics, pair digests)" % (seq_bitwise, seq_bitwise))
    print("fuzz: %d trials, %d pairs bit-exact, %d poses checked: %d in the strict class (bar 1e-4 against the sequential fp32 oracle), %d in the envelope class (of which %d because "
          "the two summation orders' pair sets part ways -- digests); %s" % (n_trials, checked_pairs, checked_poses, checked_poses - soft, soft, sets_differ, env.summary()))
    for v in env.violations:
        print("OUTSIDE THE ENVELOPE", v)
    # Round 6: no allowance by count.  Of round 5's five violators (eighteen seeds x 420 trials, 28 972 alignments) three lie INSIDE the reference's own arithmetic
    # once it is sampled at 64 perturbed starts instead of four (tests/replay_violators.py, profiles/r06/violators_replay_r06.txt); the other two are the tree
    # order's own and are NAMED in fuzz_cases.KNOWN_TREE_ORDER_DEVIATIONS with their bounds; with "sum_order" 1 all five equal the sequential oracle bit for bit.
    for v in env.ill:
        print("ILL-CONDITIONED (the reference arithmetic has no answer to 1e-2)", v)
    env.assert_only_named_exceptions()
    assert checked_pairs > 5000 and checked_poses >= 12


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


def test_randomised_aligner_structure(ctx, po):
    """Fuzz the aligner's STRUCTURE: 1-3 projective slices with their own projectors and extrinsics, Cauchy on some, an odometry
    prior on some, batches of 1-5 alignments choosing their scans through an index array, 1-12 iterations.  The split path must
    give the fused path's bits; against the oracle the first iteration has the same correspondence count and the final pose is
    within the north_star tolerance (widened only where the fp32 and fp64 oracles themselves disagree or the sets part ways)."""
    import os
    n_trials = int(os.environ.get("LSM2D_FUZZ_TRIALS", "12")); seed = int(os.environ.get("LSM2D_FUZZ_SEED", "5"))
    checked = soft = paired = sets_differ = seq_bitwise = 0
    worst_same_strict = 0.0      # largest |device - sequential-order oracle| (m or rad) in the strict class
    env = _Envelope("structure", seed)
    for spec in fuzz_cases.structure_trials(seed, n_trials):      # the draws: tests/fuzz_cases.py (shared with tests/replay_violators.py)
        trial, ns, nb, its, m, use_prior, x0, pri, all_projective = (spec[k] for k in ("trial", "ns", "nb", "its", "map", "use_prior", "x0", "pri", "all_projective"))
        al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=spec["min_inl"])
        fixed_sets, oslices, scans_per_slice = [], [], []
        for sl_spec in spec["slices"]:
            sl = fuzz_cases.structure_slice_processor(ctx, sl_spec)
            al.param_slice_processors.append(sl)
            fixed_sets.append(api.CloudSet(ctx, sl_spec["pts"], sl_spec["offs"])); scans_per_slice.append((sl_spec["pts"], sl_spec["offs"]))
            oslices.append(_oracle_slice(po, sl.slice_params()))
        mv = [api.CloudSet(ctx, m)] * ns

        def run(path, sum_order=0):
            ctx.set_option("align_path", path); ctx.set_option("sum_order", sum_order)
            try:
                return al.compute_batch(fixed_sets, mv, x0, priors=pri, want_stats=True)
            finally:
                ctx.set_option("align_path", 0); ctx.set_option("sum_order", 0)
        a = run(1)
        # Round 6: "sum_order" 1 -- pair after pair, the reference's order: EVERY alignment equals the sequential fp32 oracle `r` below bit for bit, on the
        # one-workgroup-per-alignment kernel, on whatever the library picks by itself, and on the split path
        a_seq = run(1, 1); a_seq0 = run(0, 1); a_seq2 = run(2, 1) if all_projective else None
        if all_projective:                # the split path takes projective slices only
            b = run(2)
            assert np.array_equal(a.pose, b.pose) and np.array_equal(a.information, b.information) and np.array_equal(a.status, b.status), ("split != fused", trial)
            if ns <= 2:                   # the latency kernel (k_align_pair; two slices side by side in one workgroup): the same bits, statistics included
                c = run(3)
                assert ctx.get_option("last_align_path") == 3
                assert np.array_equal(a.pose, c.pose) and np.array_equal(a.information, c.information) and np.array_equal(a.status, c.status) and \
                    np.array_equal(a.iterations, c.iterations), ("pair != fused", trial)
                for i in range(nb):
                    assert np.array_equal(a.stats[i][: a.iterations[i]], c.stats[i][: c.iterations[i]]), ("pair != fused, statistics", trial, i)
                paired += 1
        for i in range(nb):
            sc = [p[o[i]:o[i + 1]] for p, o in scans_per_slice]
            kw = dict(prior_z=pri[i][0], prior_omega=pri[i][1]) if use_prior else {}
            r = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers, **kw), oslices, sc, [m] * ns, x0[i])
            rd = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers, **kw), oslices, sc, [m] * ns, x0[i].astype(np.float64), double=True)
            rt = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers, device_order=True, **kw), oslices, sc, [m] * ns, x0[i])
            _assert_bitwise_equal_to_device_order_oracle(a, i, rt, ("trial=%d" % trial, i))
            for tag, res_s in (("fused", a_seq), ("automatic path", a_seq0), ("split", a_seq2)):
                if res_s is not None:
                    _assert_bitwise_equal_to_device_order_oracle(res_s, i, r, ("trial=%d sum_order 1, %s" % (trial, tag), i))
            seq_bitwise += 1
            assert a.stats[i]["n_correspondences"][0] == r["stats"][0].n_corr, ("first iteration", trial, i)
            dd = np.abs(r["pose"].astype(np.float64) - rd["pose"]); dd[2] = abs((dd[2] + math.pi) % (2 * math.pi) - math.pi)
            agree = r["status"] == rd["status"] == 0 and r["iterations"] == rd["iterations"] and a.status[i] == 0 and a.iterations[i] == r["iterations"]
            same_sets = bool(agree) and _same_correspondence_sets(a.stats[i], r["stats"], r["iterations"])      # exact since round 4: every iteration's pair digest
            if same_sets and 4.0 * dd.max() <= POSE_TOL_M:
                # the STRICT class: same pairs in every iteration, a well-conditioned problem -- north_star's bar against the sequential fp32 oracle
                d = np.abs(a.pose[i] - r["pose"]); d[2] = abs((d[2] + math.pi) % (2 * math.pi) - math.pi)
                assert d.max() < POSE_TOL_M, (trial, i, d, dd)
                worst_same_strict = max(worst_same_strict, float(d.max()))
                checked += 1
                continue
            # everything else -- the summation orders' pair sets part ways, the fp32 and fp64 oracles are themselves apart, a status differs -- is held to the
            # ENVELOPE of the reference's own arithmetic around the fp64 truth (round 5; the flat centimetre of rounds 3-4 is gone, and nothing is skipped)
            rr = po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers, **kw), oslices, sc, [m] * ns, x0[i], double="ref")
            verdict = env.check((trial, i, "same_sets" if same_sets else "sets_differ"), a.pose[i], int(a.status[i]), r, rd, rr,
                                perturbed=lambda: [po.align(po.aligner_params(its, min_num_inliers=al.param_min_num_inliers, **kw), oslices, sc, [m] * ns, xp) for xp in _Envelope.one_ulp_starts(x0[i])])
            sets_differ += int(bool(agree) and not same_sets)
            soft += 1
            if verdict == "violation" and os.environ.get("LSM2D_FUZZ_VERBOSE"):
                print("trial", trial, "alignment", i, dict(ns=ns, nb=nb, its=its, prior=use_prior, n_map=len(m)), env.violations[-1])
                for k_ in range(min(r["iterations"], a.iterations[i])):
                    g_ = a.stats[i][k_]; o_ = r["stats"][k_]; t_ = rd["stats"][k_] if k_ < rd["iterations"] else o_
                    print("  it %d gpu n=%d in=%d chi=%.7g | f32 n=%d in=%d chi=%.7g | f64 n=%d in=%d chi=%.7g" % (k_, g_["n_correspondences"], g_["n_inliers"], g_["chi_inliers"],
                          o_.n_corr, o_.n_in, o_.chi_in, t_.n_corr, t_.n_in, t_.chi_in))
            checked += 1
    print("structure fuzz, sum_order 1: %d of %d alignments BITWISE equal to the sequential fp32 oracle on every path (fused, automatic, split)" % (seq_bitwise, seq_bitwise))
    print("structure fuzz: %d trials, %d alignments checked: %d in the strict class (every iteration's digest equal, bar 1e-4 against the sequential fp32 oracle: largest "
          "difference %.2e), %d in the envelope class (of which %d because the two summation orders' pair sets part ways -- digests); split == fused in all, latency kernel == "
          "fused in all %d one- and two-slice trials; %s" % (n_trials, checked, checked - soft, worst_same_strict, soft, sets_differ, paired, env.summary()))
    for v in env.violations:
        print("OUTSIDE THE ENVELOPE", v)
    # Round 6: no allowance by count.  Of round 5's five violators (eighteen seeds x 420 trials, 28 972 alignments) three lie INSIDE the reference's own arithmetic
    # once it is sampled at 64 perturbed starts instead of four (tests/replay_violators.py, profiles/r06/violators_replay_r06.txt); the other two are the tree
    # order's own and are NAMED in fuzz_cases.KNOWN_TREE_ORDER_DEVIATIONS with their bounds; with "sum_order" 1 all five equal the sequential oracle bit for bit.
    for v in env.ill:
        print("ILL-CONDITIONED (the reference arithmetic has no answer to 1e-2)", v)
    env.assert_only_named_exceptions()
    assert checked >= n_trials // 2 and env.tally["status_differs"] <= max(2, checked // 50)


def test_gpu_reproduces_the_frozen_golden_bits(ctx):
    """tests/golden/oracle_regression.json holds the fp32 mirror's poses and per-iteration statistics in the kernels' summation order,
    frozen as hex floats (generated on the CPU by tests/golden/make_oracle_regression.py).  The device must give exactly those bits --
    no oracle call in this test: committed data against the HIP path."""
    g = json.load(open(golden_path("oracle_regression.json")))
    wl = synth.make_workload(3, 8000, seed=42, n_beams=361)
    finders = {"projective": lambda: api.CorrespondenceFinderProjective2f(ctx, api.PointNormal2fProjectorPolar(361, -math.pi, math.pi, 0.3, 30.0)),
               "nn": lambda: api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=0.3, normal_cos=0.8),
               "distmap": lambda: api.CorrespondenceFinderNN2D(ctx, max_distance_m=0.5, resolution=0.1, normal_cos=0.8)}
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    for name, mk in finders.items():
        al = api.MultiAligner2D(ctx, max_iterations=10, min_num_inliers=10)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(mk(), min_num_correspondences=10))
        res = al.compute_batch([fixed], [moving], wl.x0, want_stats=True)
        for c in g["cases"]:
            i = c["index"]; w = c[name + "_fp32"]["device_order"]
            assert int(res.status[i]) == w["status"], (name, i)
            assert [float(v).hex() for v in res.pose[i]] == w["pose_hex"], (name, i, res.pose[i].tolist())
            k = len(w["n_corr"])
            assert res.stats[i]["n_correspondences"][:k].tolist() == w["n_corr"]
            assert [float(v).hex() for v in res.stats[i]["chi_inliers"][:k]] == w["chi_in_hex"]


def test_gpu_reproduces_the_frozen_tracker_chain(ctx):
    """tests/golden/tracker_chain.json (digests written by the oracle on the CPU box): the HIP path, fed the same raw ranges, must
    produce the same preprocessed scans, clipped scenes, poses, information matrices and local maps at every step -- committed data
    against the device, no oracle call; once with kernel timing (every call launches at once) and once without (deferred launches,
    both scans preprocessed by one launch)."""
    import tracker_chain
    g = json.load(open(golden_path("tracker_chain.json")))
    assert tracker_chain.run_device(api, ctx, len(g["steps"])) == g["steps"]
    quiet = api.Context(0, kernel_timing=False)
    try:
        assert tracker_chain.run_device(api, quiet, len(g["steps"])) == g["steps"]
    finally:
        quiet.close()


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


# ---- the reference's own KD-tree on the device (LSM2D_FINDER_KDTREE; registration/correspondence_finder_kd_tree_2d.cpp:5-38, .h:23-34) --------
def _kd_finder(ctx, md, leaf_range=1e-2, leaf_points=20, normal_cos=0.8):
    return api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=md, normal_cos=normal_cos, max_leaf_range=leaf_range, min_leaf_points=leaf_points, search="kdtree")


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


def _kd_aligner(ctx, md=0.5, its=20, leaf_range=1e-2, leaf_points=20, robustifier=None):
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(_kd_finder(ctx, md, leaf_range, leaf_points), robustifier=robustifier, min_num_correspondences=10))
    return al


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


def _neg_eps(ctx, fixed, moving, wl):
    al = api.MultiAligner2D(ctx, max_iterations=5, termination_chi_epsilon=-1.0)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector())))
    return al.compute_batch([fixed], [moving], wl.x0)


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


def test_sweep_replication_paths_peer_same_device_and_host(ctx, small_workload):
    """lsm2d_sweep_*: how a replica gets onto its device -- device to device on one card (the rehearsal), over the fabric where
    hipDeviceCanAccessPeer allows it, from the caller's host buffer otherwise ("peer_copy" 1 forces that path: what a node without peer
    access gets) -- never changes a result.  With two or more GPUs visible the same runs on DISTINCT devices (skipped on a one-GPU box)."""
    import ctypes as C
    import torch
    from srrg2_laser_slam_2d_amd import _capi
    lib = _capi.load(); wl = small_workload
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    n = len(wl.x0); its = 8
    ap = _capi.AlignerParams(its, 10, 0.0, 0.0)
    sp = api.make_slice_params(projector=_projector(), robustifier=0, min_num_correspondences=10)
    x0 = np.ascontiguousarray(wl.x0, np.float32)
    scans = np.ascontiguousarray(wl.scan_points); offs = np.ascontiguousarray(wl.scan_offsets, np.int32); mp = np.ascontiguousarray(wl.map_points)
    want = _aligner(ctx, its=its).compute_batch([api.CloudSet(ctx, scans, offs)], [api.CloudSet(ctx, mp)], x0)
    opt = lambda sw, key: (lambda v: (lib.lsm2d_sweep_get_option(sw, key, C.byref(v)), v.value)[1])(C.c_int64(-1))
    device_sets = [[0, 0, 0]]
    if torch.cuda.device_count() >= 2:
        device_sets.append([0, 1] + ([2] if torch.cuda.device_count() >= 3 else []))
    for devices in device_sets:
        for peer_copy in (0, 1):
            sw = C.c_void_p()
            assert lib.lsm2d_sweep_create((C.c_int32 * len(devices))(*devices), len(devices), C.byref(sw)) == 0
            try:
                assert lib.lsm2d_sweep_set_option(sw, b"peer_copy", 7) == _capi.BAD_ARGUMENT and lib.lsm2d_sweep_set_option(sw, b"nonsense", 0) == _capi.BAD_ARGUMENT
                assert lib.lsm2d_sweep_set_option(sw, b"peer_copy", peer_copy) == 0 and opt(sw, b"peer_copy") == peer_copy
                assert lib.lsm2d_sweep_set_scans(sw, P(scans), P(offs), n) == 0 and lib.lsm2d_sweep_set_map(sw, P(mp), len(mp)) == 0
                by_peer, host, same = opt(sw, b"replicas_by_peer_copy"), opt(sw, b"replicas_through_host"), opt(sw, b"replicas_same_device")
                assert by_peer + host + same == len(devices) - 1
                if peer_copy == 1:
                    assert host == len(devices) - 1
                elif len(set(devices)) == 1:
                    assert same == len(devices) - 1
                pose = np.zeros((n, 3), np.float32); status = np.full(n, -7, np.int32); iters = np.zeros(n, np.int32)
                assert lib.lsm2d_sweep_align(sw, C.byref(ap), C.byref(sp), n, None, P(x0), P(pose), None, P(status), P(iters), None) == 0
                assert np.array_equal(pose, want.pose) and np.array_equal(status, want.status) and np.array_equal(iters, want.iterations), (devices, peer_copy)
            finally:
                lib.lsm2d_sweep_destroy(sw)


# ---- BASELINE configs[2] and configs[4] at their stated sizes ---------------------------------------------------------------------------
def test_configs2_full_replay_1000_steps_against_committed_digests(ctx):
    """BASELINE configs[2] at its size (usage contract apps/visual_test_tracker_2d.cpp:167-183): 1 000 tracker steps with the
    MULTI parameters -- 721-column projectors, 10 iterations, two WithSensor laser slices (Cauchy 0.01 / none) plus the odometry prior,
    raw ranges in, preprocess, clip, align, merge, everything chained on the DEVICE's own state (asynchronous clip / merge: one
    synchronisation per step, for the pose).  tests/golden/tracker_replay_1000.json holds the oracle's digests of every 50th step (scans,
    clipped scene, pose bits, information matrix, local map), written on the CPU box by tests/golden/make_tracker_chain.py and re-checked
    against the oracle by tests/test_oracle.py: a single flipped bit anywhere in the 1 000 steps changes every later digest."""
    import time
    import tracker_chain
    g = json.load(open(golden_path("tracker_replay_1000.json")))
    assert g["steps_total"] == 1000 and g["record_every"] == 50 and len(g["steps"]) == 20
    quiet = api.Context(0, kernel_timing=False)
    try:
        t0 = time.perf_counter()
        got = tracker_chain.run_device(api, quiet, 1000, record_every=50, map_capacity=60000)
        dt = time.perf_counter() - t0
    finally:
        quiet.close()
    assert [r["step"] for r in got] == [r["step"] for r in g["steps"]]
    for a, b in zip(got, g["steps"]):
        assert a == b, (a["step"], {k: (a[k], b[k]) for k in b if a[k] != b[k]})
    assert all(r["status"] == 0 for r in got) and got[-1]["map_points"] > 4000
    print("configs[2] replay: 1000 steps in %.2f s (Python driver, ranges in -> pose out, %.3f ms per step incl. the digests' downloads)" % (dt, dt))


def test_configs4_full_size_properties_1000_scans_vs_1m_map(ctx, po):
    """BASELINE configs[4] at its size: 1 000 scans x 1M-point map x 20 iterations -- too slow for the scalar oracle as a unit test beyond a
    few alignments, so: convergence to the generating pose on noise-free data (1e-4 m / 1e-4 rad), run-to-run bitwise determinism,
    permutation equivariance through the index array, the culled and the un-culled stream bit for bit, and three sampled alignments
    bitwise against the device-order oracle."""
    wl = synth.make_workload(1000, 1000000, seed=4)
    al = _aligner(ctx)
    fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
    res = al.compute_batch([fixed], [moving], wl.x0)
    assert np.all(res.status == 0)
    d = np.abs(res.pose - wl.x_true)
    assert d[:, :2].max() < POSE_TOL_M and d[:, 2].max() < POSE_TOL_RAD
    again = al.compute_batch([fixed], [moving], wl.x0)
    assert np.array_equal(again.pose, res.pose) and np.array_equal(again.information, res.information)
    perm = np.argsort(synth.Stream(11).uniform(1000)).astype(np.int32)
    res_p = al.compute_batch([fixed], [moving], wl.x0[perm], fixed_index=perm[None, :])
    assert np.array_equal(res_p.pose, res.pose[perm]) and np.array_equal(res_p.information, res.information[perm])
    ctx.set_option("cull", 0)
    try:
        plain = al.compute_batch([fixed], [moving], wl.x0)
    finally:
        ctx.set_option("cull", 1)
    assert np.array_equal(plain.pose, res.pose) and np.array_equal(plain.information, res.information) and np.array_equal(plain.iterations, res.iterations)
    # round 5 (experiments build; measured and not shipped, DESIGN App. A): the XCD lockstep -- the workgroups of an XCD walk the map in step, pass by pass
    # ("xcd_lockstep" k: nobody starts a pass before everybody on its XCD has finished the pass k - 1 back) -- changes WHEN a map point is visited, never a result:
    # with a termination criterion that ends alignments at different iterations (workgroups that go early) and with start poses that fail at once (workgroups
    # that are gone before the others have started)
    if has_experiments(ctx):
        al_eps = _aligner(ctx); al_eps.param_termination_chi_epsilon = 1e-3
        x_bad = wl.x0.copy(); x_bad[::9, 0] += 400.0
        for al_w, x0_w in ((al, wl.x0), (al_eps, wl.x0), (al, x_bad)):
            got = {}
            for w in (0, 1, 3):
                try:
                    xset(ctx, xcd_lockstep=w)
                    got[w] = al_w.compute_batch([fixed], [moving], x0_w, want_stats=True)
                    assert ctx.get_option("last_xcd_lockstep") == w
                finally:
                    xset(ctx, xcd_lockstep=0)
            for w in (1, 3):
                assert np.array_equal(got[w].pose, got[0].pose) and np.array_equal(got[w].information, got[0].information) and np.array_equal(got[w].status, got[0].status), w
                assert np.array_equal(got[w].iterations, got[0].iterations) and np.array_equal(got[w].stats, got[0].stats), w
        assert (got[0].status[::9] != 0).all() and (got[0].status == 0).sum() > 800
    for i in (0, 499, 999):
        sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        rt = po.align(po.aligner_params(20, device_order=True), [po.slice_params()], [sc], [wl.map_points], wl.x0[i])
        assert np.array_equal(res.pose[i], rt["pose"]) and np.array_equal(res.information[i], rt["H"]), i


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
