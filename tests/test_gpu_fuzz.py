"""GPU parity tests, the randomised tests: parameters, structure (both also with sum_order 1, bitwise against the sequential oracle); tools/fuzz_soak.sh runs them over many seeds.

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
