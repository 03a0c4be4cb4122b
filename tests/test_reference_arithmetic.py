"""The oracle's reference-arithmetic mode `_r` (libm atan2f / sinf / cosf / logf, no FMA, Eigen's association; oracle/lsm2d_oracle.h)
against the fp32 mirror `_f` that the HIP kernels reproduce bit for bit, and the believed upstream KD-tree descent against the exact
nearest-neighbour search.  CPU only; the full table is produced by tests/parity_study.py (PARITY.md section 5).

What is asserted: on every BASELINE configuration the two arithmetics end within the north_star tolerance (1e-4 m / 1e-4 rad) of
each other -- on clean and on noisy data -- although a fraction of a percent of the z-buffer winners and pairs differ; the
approximate tree changes several percent of the pairs and, on these inputs, none of the poses beyond 1e-4."""
import math

import numpy as np
import pytest

import parity_study as ps
from srrg2_laser_slam_2d_amd import synth


@pytest.fixture(scope="module")
def study():
    return ps.run(quick=True)


def test_reference_arithmetic_stays_within_the_pose_tolerance_on_every_config(study):
    rows, _ = study
    assert len(rows) == 6
    for r in rows:
        assert r["dpose_m"] < 1e-4 and r["dpose_rad"] < 1e-4, r
        assert r["winners_x0"] < 0.03 and r["pairs_x0"] < 0.06, r          # a few columns flip at bin edges, nothing more
    for r in rows[:5]:                                                     # clean data: both modes sit on the generating pose
        assert r["err_f_vs_truth"] < 1e-4 and r["err_r_vs_truth"] < 1e-4, r
    # the dense maps are where near-ties live: the 100k / 1M maps do show differing winners (the study measures something)
    assert rows[1]["winners_x0"] > 0 and rows[4]["winners_x0"] > rows[0]["winners_x0"]


def test_believed_upstream_kdtree_is_approximate_but_ends_on_the_same_pose(study):
    _, kd = study
    for r in kd:
        assert 0.005 < r["pairs_x0"] < 0.3, r            # the single-leaf descent does miss nearest neighbours ...
        assert r["pairs_kdtree"] <= r["pairs_exact"]      # ... and a miss can only lose a pair (the leaf's best is farther or absent)
        assert r["dpose_m"] < 1e-4 and r["dpose_rad"] < 1e-4, r
        assert r["err_kdtree_vs_truth"] < 1e-4, r


def test_kdtree_oracle_semantics(po):
    """Hand-checkable cases of the restated tree: a single leaf is exact; the -1 / max_distance convention of
    correspondence_finder_kd_tree_2d.cpp:18-21; a query next to a splitting plane misses the nearer point on the other side."""
    rng = np.random.default_rng(1)
    # (1) fewer points than min_leaf_points: one leaf, so the search is the brute-force one
    fixed = np.zeros((12, 4), np.float32); fixed[:, :2] = rng.uniform(-1, 1, (12, 2)); fixed[:, 2] = 1.0
    moving = np.zeros((30, 4), np.float32); moving[:, :2] = rng.uniform(-1, 1, (30, 2)); moving[:, 2] = 1.0
    ident = np.zeros(3, np.float32)
    sp_kd = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.4, normal_cos=-1.0)
    sp_nn = po.slice_params(finder=po.FINDER_NN, max_distance=0.4, normal_cos=-1.0)
    for mode in (False, True, "ref"):
        a = po.find(sp_kd, fixed, moving, ident, double=mode); b = po.find(sp_nn, fixed, moving, ident, double=mode, brute=(mode is False))
        assert np.array_equal(a, b)
    # (2) two clusters split by the plane x = 0 (principal axis x): the plane passes through the mean (x ~ -0.07): a query at x = -0.1 descends left and never sees the point at x = +0.02
    left = np.stack([np.linspace(-1.0, -0.3, 15), np.linspace(-0.2, 0.2, 15)], 1)
    right = np.stack([np.linspace(0.02, 1.0, 15), np.linspace(-0.2, 0.2, 15)], 1)
    fixed = np.zeros((30, 4), np.float32); fixed[:, :2] = np.concatenate([left, right]); fixed[:, 2] = 1.0
    q = np.zeros((1, 4), np.float32); q[0, :2] = (-0.1, -0.2); q[0, 2] = 1.0
    sp_kd = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=0.5, normal_cos=-1.0, kd_min_leaf_points=20)
    sp_nn = po.slice_params(finder=po.FINDER_NN, max_distance=0.5, normal_cos=-1.0)
    exact = po.find(sp_nn, fixed, q, ident, double="ref"); approx = po.find(sp_kd, fixed, q, ident, double="ref")
    assert exact.tolist() == [[15, 0]]                       # the true neighbour is the first point of the right cluster
    assert approx.tolist() in ([], [[14, 0]])                # the left leaf: its nearest point is 0.45 m away (or beyond max_distance)
    # (3) nothing within max_distance -> no pair
    far = q.copy(); far[0, :2] = (50.0, 50.0)
    assert len(po.find(sp_kd, fixed, far, ident, double="ref")) == 0


def test_reference_mode_uses_libm_and_differs_from_the_mirror_only_in_the_last_bits(po):
    """The `_r` projector calls atan2f: its columns equal floor(K00 * atan2f + K01) evaluated with numpy's float32 libm path, and its
    depths / transformed points are within a few ulp of the mirror's."""
    wl = synth.make_workload(1, 20000, seed=9)
    pr = po.Projector(1081, -math.pi, math.pi, 0.3, 30.0, 0.0)
    src_f, dep_f, xyn_f = po.project(pr, wl.map_points, wl.x0[0])
    src_r, dep_r, xyn_r = po.project(pr, wl.map_points, wl.x0[0], double="ref")
    same = (src_f == src_r) & (src_f >= 0)
    assert same.sum() > 0.97 * (src_f >= 0).sum() and (src_f >= 0).sum() > 900
    assert np.allclose(dep_f[same], dep_r[same], rtol=2e-6, atol=0) and np.allclose(xyn_f[same], xyn_r[same], atol=2e-5)
    assert not np.array_equal(xyn_f[same], xyn_r[same])      # ... but not the same bits: FMA vs separate roundings
