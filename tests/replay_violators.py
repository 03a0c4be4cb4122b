#!/usr/bin/env python3
"""Replay, on the CPU alone, the alignments the round-5 fuzz soak found OUTSIDE its envelope (profiles/r05/fuzz_soak_r05r_*.log), with a denser sample of the
reference's own arithmetic around each of them:

    python tests/replay_violators.py [--samples 64] > profiles/r06/violators_replay_r06.txt

The device equals the oracle's device-order mirror bit for bit in every alignment the GPU suite has ever checked, so "what the device did" is reproduced here by
that mirror (lsmo_align_f with device_order = 1) -- the logged device poses are compared with it below, to the bit.  For every case:

  * the fp64 oracle (the truth the envelope is centred on), the sequential fp32 oracle `_f` (the reference's order: what the device computes with "sum_order" 1),
    the reference-arithmetic oracle `_r` (libm, no FMA, pair after pair) and the device-order mirror (the default "sum_order" 0: tree sums), all from the logged start pose;
  * the same three fp32 evaluations from N start poses moved by one or two ulps per component (all 26 one-ulp patterns of {-1, 0, +1}^3 first, then two-ulp
    patterns from a seeded stream) -- what another compiler's, or another summation order's, last bit does to each arithmetic;
  * per arithmetic: how many of the N runs end more than 1e-4 m / 1e-4 rad from the fp64 oracle, the largest and the median distance.

Reading: if the sequential / reference arithmetics NEVER leave 1e-4 in N runs while the tree order does, the deviation is the tree order's own (it picks another pair
at a gate); if they do too, the alignment sits on a gate for every arithmetic and the soak merely sampled the reference too thinly (four runs).
Test infrastructure: imports the oracle; nothing here is product code."""
from __future__ import annotations

import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import fuzz_cases  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

# (test, seed, trial, alignment, logged device pose): profiles/r05/fuzz_soak_r05r_14_seeds.log, ..._4_more_seeds.log
CASES = [
    ("structure", 7, 29, 3, [-19.815784454345703, -4.790680408477783, 0.8985986709594727], "OUTSIDE"),
    ("structure", 42, 97, 1, [-4.756843566894531, 15.005985260009766, 2.266704797744751], "OUTSIDE"),
    ("parameters", 2024, 237, 0, [-17.688289642333984, 5.223707675933838, -2.73071026802063], "OUTSIDE"),
    ("structure", 17, 346, 0, [-19.81760025024414, -4.79147481918335, 0.8986247181892395], "OUTSIDE"),
    ("structure", 8675309, 66, 1, [1.8744546175003052, -18.117185592651367, -0.9545098543167114], "OUTSIDE"),
    ("parameters", 4711, 219, 0, None, "ILL-CONDITIONED"),
    ("parameters", 7, 25, 0, None, "ILL-CONDITIONED (round 6: decided before the perturbed runs may widen the envelope)"),
    ("parameters", 42, 30, 0, None, "ILL-CONDITIONED (round 6)"),
]


def pose_diff(p, q):
    d = np.abs(np.asarray(p, np.float64) - np.asarray(q, np.float64)); d[2] = abs((d[2] + math.pi) % (2 * math.pi) - math.pi)
    return float(d[:2].max()), float(d[2])


perturbed_starts = fuzz_cases.perturbed_starts      # the same sample the GPU fuzz tests' envelope uses


def inputs_of(test, seed, trial, alignment):
    if test == "structure":
        for spec in fuzz_cases.structure_trials(seed, trial + 1):
            pass
        assert spec["trial"] == trial
        osl, fx, mv, kw = fuzz_cases.structure_oracle_inputs(po, spec, alignment)
        return osl, fx, mv, kw, spec["x0"][alignment], dict(ns=spec["ns"], its=spec["its"], prior=spec["use_prior"], n_map=len(spec["map"]),
                                                           finders=[sl["kind"] for sl in spec["slices"]], cauchy=[sl["cauchy"] for sl in spec["slices"]])
    for spec in fuzz_cases.parameter_trials(seed, trial + 1):
        pass
    assert spec["trial"] == trial
    _, osp = fuzz_cases.parameter_finder(None, spec)
    fuzz_cases.parameter_aligner_slice(po, spec, osp)
    return [osp], [spec["scan"]], [spec["map"]], dict(max_iterations=spec["its"], min_num_inliers=spec["min_inl"]), spec["x0"], \
        dict(ns=1, its=spec["its"], prior=False, n_map=spec["n_map"], finders=[spec["finder"]], cauchy=[spec["cauchy"]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=64)
    args = ap.parse_args()
    po.lib()
    print("# replay of the round-5 soak's violators with %d perturbed start poses each (26 one-ulp patterns, then two-ulp ones); distances are to the fp64 oracle run from the" % args.samples)
    print("# LOGGED start pose, metres / radians; 'over' = runs ending more than 1e-4 m or 1e-4 rad away")
    summary = []
    for test, seed, trial, ali, logged, kind in CASES:
        osl, fx, mv, kw, x0, desc = inputs_of(test, seed, trial, ali)

        def run(x, **more):
            double = more.pop("double", False)
            return po.align(po.aligner_params(**kw, **more), osl, fx, mv, np.asarray(x, np.float64) if double is True else x, double=double)
        rd = run(x0, double=True); r = run(x0); rr = run(x0, double="ref"); rt = run(x0, device_order=True)
        print("\n== %s fuzz, seed %d, trial %d, alignment %d (%s in round 5): %s" % (test, seed, trial, ali, kind, desc))
        print("   start pose %s" % np.asarray(x0).tolist())
        if logged is not None:
            same = np.array_equal(np.asarray(logged, np.float32), rt["pose"])
            print("   logged device pose == device-order mirror here, bit for bit: %s" % same)
            assert same, (logged, rt["pose"].tolist())
        print("   fp64 oracle            status %d its %2d pose %s" % (rd["status"], rd["iterations"], rd["pose"].tolist()))
        for name, o in (("sequential fp32 (_f)", r), ("reference arith. (_r)", rr), ("device order (trees)", rt)):
            dm, dr = pose_diff(o["pose"], rd["pose"])
            print("   %-22s status %d its %2d  |pose - fp64| = %.3e m %.3e rad" % (name, o["status"], o["iterations"], dm, dr))
        rows = {"sequential fp32 (_f)": [], "reference arith. (_r)": [], "device order (trees)": []}
        for steps, x in perturbed_starts(x0, args.samples):
            for name, o in (("sequential fp32 (_f)", run(x)), ("reference arith. (_r)", run(x, double="ref")), ("device order (trees)", run(x, device_order=True))):
                rows[name].append(pose_diff(o["pose"], rd["pose"]) + (o["status"],))
        line = {}
        for name, v in rows.items():
            ok = [(a, b) for a, b, st in v if st == 0]
            over = sum(1 for a, b in ok if a > 1e-4 or b > 1e-4)
            dm = [a for a, _ in ok]; dr = [b for _, b in ok]
            print("   %d perturbed starts, %-22s %2d over, %2d failed; max %.3e m %.3e rad; median %.3e m %.3e rad" % (
                len(v), name + ":", over, len(v) - len(ok), max(dm, default=0.0), max(dr, default=0.0), float(np.median(dm)) if dm else 0.0, float(np.median(dr)) if dr else 0.0))
            line[name] = (over, len(v), max(dm, default=0.0))
        summary.append((test, seed, trial, ali, kind, line))
    print("\n# summary: runs over 1e-4 of N (largest distance in metres)")
    print("# %-44s %-22s %-22s %-22s" % ("case", "sequential fp32", "reference arithmetic", "device order (trees)"))
    for test, seed, trial, ali, kind, line in summary:
        cells = ["%d / %d (%.2e)" % line[k] for k in ("sequential fp32 (_f)", "reference arith. (_r)", "device order (trees)")]
        print("# %-44s %-22s %-22s %-22s" % ("%s seed %d trial %d #%d %s" % (test, seed, trial, ali, kind), *cells))


if __name__ == "__main__":
    main()
