"""Shared helpers of the GPU parity tests (tests/test_gpu_*.py): the bars, the bitwise comparison with the oracle, the envelope of the reference's own arithmetic
that the fuzz tests hold the default summation order to, small constructors.  Test infrastructure: imports the oracle through the `po` fixture only."""
import json
import math

import numpy as np
import pytest

import fuzz_cases
from conftest import golden_path, has_experiments, need_experiments, xset
from srrg2_laser_slam_2d_amd import api, synth

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
        # (an alignment whose two reference-arithmetic evaluations are THEMSELVES a hundred bars from the fp64 oracle is ill-conditioned: decided before their
        # distance may serve as a yardstick -- round 6; round 5 passed such runaways, up to metres, as "within 3 x the reference arithmetic's own distance")
        if (em <= self.ILL and er <= self.ILL) and dm <= max(POSE_TOL_M, 3.0 * em) and dr <= max(POSE_TOL_RAD, 3.0 * er):
            self.tally["needs_factor"] += 1; self.worst["needs_factor"] = max(self.worst["needs_factor"], dm, dr); return "needs_factor"
        pm = pr_ = 0.0
        if perturbed is not None and em <= self.ILL and er <= self.ILL:
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


def _aligner(ctx, cols=1081, its=20, **slice_kw):
    finder = api.CorrespondenceFinderProjective2f(ctx, _projector(cols))
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10, **slice_kw))
    return al


def _nn_aligner(ctx, md=0.5, its=20):
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=md), min_num_correspondences=10))
    return al


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


# ---- the reference's own KD-tree on the device (LSM2D_FINDER_KDTREE; registration/correspondence_finder_kd_tree_2d.cpp:5-38, .h:23-34) --------
def _kd_finder(ctx, md, leaf_range=1e-2, leaf_points=20, normal_cos=0.8):
    return api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=md, normal_cos=normal_cos, max_leaf_range=leaf_range, min_leaf_points=leaf_points, search="kdtree")


def _kd_aligner(ctx, md=0.5, its=20, leaf_range=1e-2, leaf_points=20, robustifier=None):
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(_kd_finder(ctx, md, leaf_range, leaf_points), robustifier=robustifier, min_num_correspondences=10))
    return al


def _neg_eps(ctx, fixed, moving, wl):
    al = api.MultiAligner2D(ctx, max_iterations=5, termination_chi_epsilon=-1.0)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, _projector())))
    return al.compute_batch([fixed], [moving], wl.x0)
