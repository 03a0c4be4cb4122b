"""How far is the fixed-polynomial arithmetic of the HIP path from the reference's own arithmetic?  (CPU study, oracle only.)

The HIP kernels are bit-identical to the oracle's fp32 mirror `_f` (proved on the GPU by tests/test_gpu_*.py); `_f` evaluates
atan2 / sin / cos / log as fixed polynomials and fuses multiply-adds.  The reference calls libm and has no FMA.  This script runs
both arithmetics (`_f` and the reference-arithmetic mode `_r` of oracle/lsm2d_oracle.h) on every BASELINE.json configuration and
reports, per configuration:
  winners_x0   fraction of z-buffer columns whose winning moving point differs at the initial pose
  pairs_x0     fraction of correspondence pairs that differ at the initial pose (symmetric difference / union)
  pairs_it     the same per iteration, both finders evaluated at the SAME poses (the `_r` trajectory), mean and max over iterations
  dpose        max |pose_f - pose_r| after all iterations (m, rad) -- the bar is 1e-4 / 1e-4 (BASELINE.json north_star)
  vs truth     max error of either mode against the generating pose
and for the NN finder: exact search vs the believed upstream single-leaf KD-tree descent.

    python tests/parity_study.py [--quick] [--json out.json]      (prints the markdown table of PARITY.md section 5)
"""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po                      # noqa: E402
from srrg2_laser_slam_2d_amd import synth             # noqa: E402


def _pairs_diff(a, b):
    sa = {tuple(r) for r in a.tolist()}; sb = {tuple(r) for r in b.tolist()}
    u = len(sa | sb)
    return (len(sa ^ sb) / u) if u else 0.0


def _ang(d):
    return np.abs((d + np.pi) % (2 * np.pi) - np.pi)


def study_projective(name, n_map, n_samples, iters, seed, cols=1081, range_max=30.0, normal_cos=0.8, point_distance=0.5,
                     tau=0.0, lockstep=True, noise=0.0):
    world = synth.make_world(seed)
    wl = synth.make_workload(n_samples, n_map, seed=seed, world=world, map_noise=noise, scan_noise=noise)
    kw = dict(canvas_cols=cols, range_max=range_max, normal_cos=normal_cos, point_distance=point_distance)
    if tau > 0:
        kw.update(robustifier=po.ROBUST_CAUCHY, chi_threshold=tau)
    sp = po.slice_params(**kw)
    win, px0, pit_mean, pit_max, dxy, dth, ef, er = [], [], [], [], [], [], [], []
    for i in range(n_samples):
        sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        x0 = wl.x0[i]
        src_f, _, _ = po.project(sp.projector, wl.map_points, x0)
        src_r, _, _ = po.project(sp.projector, wl.map_points, x0, double="ref")
        both = (src_f >= 0) | (src_r >= 0)
        win.append(float((src_f != src_r)[both].mean()) if both.any() else 0.0)
        px0.append(_pairs_diff(po.find(sp, sc, wl.map_points, x0), po.find(sp, sc, wl.map_points, x0, double="ref")))
        rf = po.align(po.aligner_params(iters), [sp], [sc], [wl.map_points], x0)
        rr = po.align(po.aligner_params(iters), [sp], [sc], [wl.map_points], x0, double="ref")
        assert rf["status"] == 0 and rr["status"] == 0, (name, i, rf["status"], rr["status"])
        d = np.abs(rf["pose"] - rr["pose"]); dxy.append(float(d[:2].max())); dth.append(float(_ang(d[2])))
        t = wl.x_true[i]
        ef.append(max(float(np.abs(rf["pose"][:2] - t[:2]).max()), float(_ang(rf["pose"][2] - t[2]))))
        er.append(max(float(np.abs(rr["pose"][:2] - t[:2]).max()), float(_ang(rr["pose"][2] - t[2]))))
        if lockstep:
            fr = []
            for k in range(1, iters):        # pose after k iterations of the reference-arithmetic run
                xk = po.align(po.aligner_params(k), [sp], [sc], [wl.map_points], x0, double="ref")["pose"]
                fr.append(_pairs_diff(po.find(sp, sc, wl.map_points, xk), po.find(sp, sc, wl.map_points, xk, double="ref")))
            pit_mean.append(float(np.mean(fr))); pit_max.append(float(np.max(fr)))
    return dict(config=name, finder="projective", samples=n_samples, iterations=iters,
                winners_x0=float(np.mean(win)), winners_x0_max=float(np.max(win)), pairs_x0=float(np.mean(px0)),
                pairs_it_mean=float(np.mean(pit_mean)) if pit_mean else None, pairs_it_max=float(np.max(pit_max)) if pit_max else None,
                dpose_m=float(np.max(dxy)), dpose_rad=float(np.max(dth)), err_f_vs_truth=float(np.max(ef)), err_r_vs_truth=float(np.max(er)))


def study_kdtree(name, n_map, n_samples, iters, seed, role, max_distance=0.5):
    world = synth.make_world(seed)
    wl = synth.make_workload(n_samples, n_map, seed=seed, world=world)
    sp_exact = po.slice_params(finder=po.FINDER_NN, max_distance=max_distance)
    sp_kd = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=max_distance)
    md, dxy, dth, ek, ee, nk, ne = [], [], [], [], [], [], []
    for i in range(n_samples):
        sc = wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]
        if role == "A":
            fixed, moving, x0, t = sc, wl.map_points, wl.x0[i], wl.x_true[i]
        else:
            fixed, moving = wl.map_points, sc
            x0 = synth.invert_poses(wl.x0[i:i + 1].astype(np.float64))[0].astype(np.float32); t = synth.invert_poses(wl.x_true[i:i + 1])[0]
        pe = po.find(sp_exact, fixed, moving, x0, double="ref"); pk = po.find(sp_kd, fixed, moving, x0, double="ref")
        md.append(_pairs_diff(pe, pk)); ne.append(len(pe)); nk.append(len(pk))
        re_ = po.align(po.aligner_params(iters), [sp_exact], [fixed], [moving], x0, double="ref")
        rk = po.align(po.aligner_params(iters), [sp_kd], [fixed], [moving], x0, double="ref")
        d = np.abs(re_["pose"] - rk["pose"]); dxy.append(float(d[:2].max())); dth.append(float(_ang(d[2])))
        ee.append(max(float(np.abs(re_["pose"][:2] - t[:2]).max()), float(_ang(re_["pose"][2] - t[2]))))
        ek.append(max(float(np.abs(rk["pose"][:2] - t[:2]).max()), float(_ang(rk["pose"][2] - t[2]))))
    return dict(config=name, finder="NN exact vs believed upstream KD-tree (single-leaf descent), role " + role, samples=n_samples, iterations=iters,
                pairs_x0=float(np.mean(md)), pairs_exact=float(np.mean(ne)), pairs_kdtree=float(np.mean(nk)),
                dpose_m=float(np.max(dxy)), dpose_rad=float(np.max(dth)), err_exact_vs_truth=float(np.max(ee)), err_kdtree_vs_truth=float(np.max(ek)))


def run(quick=False):
    s = (lambda full, q: q if quick else full)
    rows = [
        study_projective("configs[0] 1 scan vs 10k map", 10000, s(16, 4), 20, 11),
        study_projective("configs[1] scans vs 100k map", 100000, s(32, 4), 20, 0),
        study_projective("configs[2] MULTI tracker slice (721 cols, 20 m, cos 0.9, Cauchy 0.01, 10 it)", 5000, s(32, 4), 10, 7, cols=721,
                         range_max=20.0, normal_cos=0.9, tau=0.01),
        study_projective("configs[3] loop closure (100k submap, Cauchy 0.05)", 100000, s(32, 4), 20, 3, tau=0.05),
        study_projective("configs[4] scans vs 1M map", 1000000, s(8, 2), 20, 5, lockstep=not quick),
        study_projective("configs[1] with sigma = 1 cm noise on map and scans (truth columns: distance to the generating pose, not an error bar)",
                         100000, s(32, 4), 20, 0, noise=0.01),
    ]
    kd = [
        study_kdtree("configs[1] role B (tree over the 100k map, scan queries)", 100000, s(16, 3), 20, 0, "B"),
        study_kdtree("configs[1] role A (tree over the scan, map queries)", 20000, s(8, 2), 20, 0, "A"),
    ]
    return rows, kd


def markdown(rows, kd):
    out = ["| configuration | samples | z-buffer winners differing at X0 (mean / worst scan) | pairs differing at X0 | pairs differing per iteration, same poses (mean / worst) | max pose delta `_f` vs `_r` | `_f` vs truth | `_r` vs truth |",
           "|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| %s | %d | %.3f %% / %.3f %% | %.3f %% | %s | %.1e m / %.1e rad | %.1e | %.1e |" % (
            r["config"], r["samples"], 100 * r["winners_x0"], 100 * r["winners_x0_max"], 100 * r["pairs_x0"],
            ("%.3f %% / %.3f %%" % (100 * r["pairs_it_mean"], 100 * r["pairs_it_max"])) if r["pairs_it_mean"] is not None else "n/a",
            r["dpose_m"], r["dpose_rad"], r["err_f_vs_truth"], r["err_r_vs_truth"]))
    out += ["", "| NN finder: exact search vs believed upstream KD-tree | samples | pairs differing at X0 | pairs exact / kd-tree | max pose delta | exact vs truth | kd-tree vs truth |",
            "|---|---|---|---|---|---|---|"]
    for r in kd:
        out.append("| %s | %d | %.2f %% | %.0f / %.0f | %.1e m / %.1e rad | %.1e | %.1e |" % (
            r["config"], r["samples"], 100 * r["pairs_x0"], r["pairs_exact"], r["pairs_kdtree"], r["dpose_m"], r["dpose_rad"],
            r["err_exact_vs_truth"], r["err_kdtree_vs_truth"]))
    return "\n".join(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true"); ap.add_argument("--json", default="")
    a = ap.parse_args()
    rows, kd = run(a.quick)
    print(markdown(rows, kd))
    if a.json:
        json.dump(dict(projective=rows, kdtree=kd), open(a.json, "w"), indent=1)
