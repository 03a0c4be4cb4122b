"""The randomised aligner cases of the GPU fuzz tests, as DATA: generators that draw every trial's parameters from a seeded stream and hand them over as
plain dictionaries, so that (i) the GPU tests (tests/test_gpu_fuzz.py: test_randomised_parameters_finder_and_aligner, test_randomised_aligner_structure)
build their api objects from them and (ii) tests/replay_violators.py can re-create ANY trial of ANY seed on the CPU alone -- a soak's violator is named by
(test, seed, trial, alignment) and nothing else.  The order of the draws is that of rounds 3-5's tests, so the soak logs under profiles/ still name the same cases.

Nothing here touches the GPU; the api classes are used only as parameter holders (their constructors store the context they are given)."""
from __future__ import annotations

import itertools
import math

import numpy as np

from srrg2_laser_slam_2d_amd import api, synth


def oracle_slice(po, sp):
    """oracle SliceParams with the same values as an ABI SliceParams"""
    return po.slice_params(finder=sp.finder, canvas_cols=sp.projector.canvas_cols, angle_min=sp.projector.angle_min,
                           angle_max=sp.projector.angle_max, range_min=sp.projector.range_min, range_max=sp.projector.range_max,
                           col_offset=sp.projector.col_offset, point_distance=sp.point_distance, normal_cos=sp.normal_cos,
                           max_distance=sp.max_distance, resolution=sp.resolution, robustifier=sp.robustifier,
                           chi_threshold=sp.chi_threshold, min_num_correspondences=sp.min_num_correspondences,
                           sensor_in_robot=tuple(sp.sensor_in_robot), kd_max_leaf_range=sp.kd_max_leaf_range,
                           kd_min_leaf_points=sp.kd_min_leaf_points)


# ---- test_randomised_aligner_structure ------------------------------------------------------------------------------------------------------------
def structure_trials(seed: int, n_trials: int):
    """1-3 slices with their own projectors and extrinsics, Cauchy on some, an odometry prior on some, batches of 1-5 alignments, 1-12 iterations; every fourth
    trial mixes the three finder kinds across its slices.  Yields one dict per trial."""
    rng = np.random.default_rng(seed)
    world = synth.make_world(7)
    maps = {n: synth.make_map(world, n, noise_sigma=0.003, seed=n + 3) for n in (4000, 30000)}
    poses = synth.sample_poses(world, 8, seed=17)
    for trial in range(n_trials):
        ns = int(rng.integers(1, 4)); nb = int(rng.integers(1, 6)); its = int(rng.integers(1, 13)); m = maps[(4000, 30000)[trial % 2]]
        use_prior = bool(trial % 3 == 0)
        robots = poses[rng.integers(0, 8, nb)]
        guess = synth.compose_poses(robots, rng.uniform(-0.04, 0.04, (nb, 3)))
        x0 = synth.invert_poses(guess).astype(np.float32)
        min_inl = int(rng.integers(0, 30))
        slices = []
        for s in range(ns):
            cols = int(rng.integers(200, 1300)); rmax = float(rng.uniform(8.0, 30.0)); ncos = float(rng.uniform(0.5, 0.9)); pd = float(rng.uniform(0.2, 1.0))
            S = np.float32([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-3, 3)]) if (trial + s) % 2 else np.zeros(3, np.float32)
            cauchy = bool((trial + s) % 3 == 1); tau = float(rng.uniform(0.005, 0.05)); mc = int(rng.integers(0, 20))
            kind = int(rng.integers(0, 3)) if trial % 4 == 3 else 0          # every 4th trial mixes the three finders across its slices
            md = res = 0.0
            if kind == 1:
                md = float(rng.uniform(0.1, 0.6))
            elif kind == 2:
                md = float(rng.uniform(0.2, 0.6)); res = float(rng.uniform(0.05, 0.15))
            n_beams = int(rng.integers(300, 1100))
            pts, offs = synth.make_scans(world, synth.compose_poses(robots, np.tile(S[None, :].astype(np.float64), (nb, 1))), n_beams=n_beams,
                                         noise_sigma=0.003, seed=trial * 7 + s)
            slices.append(dict(kind=kind, cols=cols, rmax=rmax, ncos=ncos, pd=pd, S=S, cauchy=cauchy, tau=tau, mc=mc, md=md, res=res, pts=pts, offs=offs))
        pri = [(np.zeros(3, np.float32) + x0[i], np.diag(rng.uniform(5.0, 80.0, 3)).astype(np.float32)) for i in range(nb)] if use_prior else None
        yield dict(trial=trial, ns=ns, nb=nb, its=its, map=m, use_prior=use_prior, x0=x0, min_inl=min_inl, slices=slices, pri=pri,
                   all_projective=all(sl["kind"] == 0 for sl in slices))


def structure_slice_processor(ctx, sl):
    """the api slice processor of one slice of a structure trial (ctx may be None: parameters only)"""
    proj = api.PointNormal2fProjectorPolar(sl["cols"], -math.pi, math.pi, 0.3, sl["rmax"])
    if sl["kind"] == 0:
        f = api.CorrespondenceFinderProjective2f(ctx, proj, sl["pd"], sl["ncos"])
    elif sl["kind"] == 1:
        f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=sl["md"], normal_cos=sl["ncos"])
    else:
        f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=sl["md"], resolution=sl["res"], normal_cos=sl["ncos"])
    rob = api.RobustifierCauchy(sl["tau"]) if sl["cauchy"] else None
    if sl["S"].any():
        return api.AlignerSliceProcessorLaser2DWithSensor(f, sensor_in_robot=sl["S"], robustifier=rob, min_num_correspondences=sl["mc"])
    return api.AlignerSliceProcessorLaser2D(f, robustifier=rob, min_num_correspondences=sl["mc"])


def structure_oracle_inputs(po, spec, i):
    """(oracle slices, fixed clouds, moving clouds, aligner keyword arguments) of alignment i of a structure trial"""
    oslices = [oracle_slice(po, structure_slice_processor(None, sl).slice_params()) for sl in spec["slices"]]
    sc = [sl["pts"][sl["offs"][i]:sl["offs"][i + 1]] for sl in spec["slices"]]
    kw = dict(prior_z=spec["pri"][i][0], prior_omega=spec["pri"][i][1]) if spec["use_prior"] else {}
    return oslices, sc, [spec["map"]] * spec["ns"], dict(max_iterations=spec["its"], min_num_inliers=spec["min_inl"], **kw)


# ---- test_randomised_parameters_finder_and_aligner ------------------------------------------------------------------------------------------------
def parameter_trials(seed: int, n_trials: int):
    """One slice, one alignment: asymmetric fields of view, odd canvas sizes, column rounding, tight and wide gates, all four finder kinds, Cauchy on/off, sensor
    extrinsics.  Yields one dict per trial."""
    rng = np.random.default_rng(seed)
    rng_kd = np.random.default_rng(seed + 1000)      # (a generator of its own: the other trials keep the parameter sequences of earlier rounds' soaks)
    world = synth.make_world(9)
    maps = {n: synth.make_map(world, n, noise_sigma=0.003, seed=n) for n in (3000, 20000)}
    poses = synth.sample_poses(world, 12, seed=3)
    for trial in range(n_trials):
        n_map = (3000, 20000)[trial % 2]
        m = maps[n_map]
        beams = int(rng.integers(90, 1200))
        scan, _ = synth.make_scans(world, poses[trial % 12:trial % 12 + 1], n_beams=beams, fov_deg=float(rng.uniform(90, 300)))
        x_true, x0 = synth.initial_guesses(poses[trial % 12:trial % 12 + 1], seed=trial, scale=float(rng.uniform(0.0, 0.08)))
        x0 = x0[0].astype(np.float32)
        finder = trial % 3 if trial % 7 else 3          # every seventh trial: the reference's own KD-tree, built on the device, random leaf parameters
        a0 = float(rng.uniform(-math.pi, -0.5)); a1 = float(rng.uniform(0.5, math.pi))
        cols = int(rng.integers(64, 2000)); off = float(rng.choice([0.0, 0.5]))
        rmin = float(rng.uniform(0.0, 1.0)); rmax = float(rng.uniform(5.0, 40.0))
        pd = float(rng.uniform(0.05, 1.5)); nc = float(rng.uniform(0.3, 0.95)); md = float(rng.uniform(0.02, 0.8)); res = float(rng.uniform(0.03, 0.2))
        cauchy = bool(trial % 4 == 1); tau = float(rng.uniform(0.005, 0.1)); mc = int(rng.integers(0, 30))
        S = (0.0, 0.0, 0.0) if trial % 5 else (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-1, 1)))
        its = int(rng.integers(1, 15)); min_inl = int(rng.integers(0, 50))
        lr, lp = 1e-2, 20
        if finder == 3:
            lr = float(10.0 ** rng_kd.uniform(-3, 0)); lp = int(rng_kd.integers(1, 60))
        yield dict(trial=trial, n_map=n_map, map=m, beams=beams, scan=scan, x0=x0, finder=finder, a0=a0, a1=a1, cols=cols, off=off, rmin=rmin, rmax=rmax, pd=pd, nc=nc,
                   md=md, res=res, cauchy=cauchy, tau=tau, mc=mc, S=S, its=its, min_inl=min_inl, lr=lr, lp=lp)


def parameter_finder(ctx, spec):
    """(api finder, oracle slice parameters of the FINDER -- no robustifier yet) of a parameter trial"""
    from oracle import pyoracle as po
    proj = api.PointNormal2fProjectorPolar(spec["cols"], spec["a0"], spec["a1"], spec["rmin"], spec["rmax"], spec["off"])
    if spec["finder"] == 0:
        f = api.CorrespondenceFinderProjective2f(ctx, proj, spec["pd"], spec["nc"])
        osp = po.slice_params(canvas_cols=spec["cols"], angle_min=spec["a0"], angle_max=spec["a1"], range_min=spec["rmin"], range_max=spec["rmax"], col_offset=spec["off"],
                              point_distance=spec["pd"], normal_cos=spec["nc"])
    elif spec["finder"] == 1:
        f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=spec["md"], normal_cos=spec["nc"])
        osp = po.slice_params(finder=po.FINDER_NN, max_distance=spec["md"], normal_cos=spec["nc"])
    elif spec["finder"] == 3:
        f = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=spec["md"], normal_cos=spec["nc"], max_leaf_range=spec["lr"], min_leaf_points=spec["lp"], search="kdtree")
        osp = po.slice_params(finder=po.FINDER_KDTREE_APPROX, max_distance=spec["md"], normal_cos=spec["nc"], kd_max_leaf_range=spec["lr"], kd_min_leaf_points=spec["lp"])
    else:
        f = api.CorrespondenceFinderNN2D(ctx, max_distance_m=spec["md"], resolution=spec["res"], normal_cos=spec["nc"])
        osp = po.slice_params(finder=po.FINDER_DISTMAP, max_distance=spec["md"], resolution=spec["res"], normal_cos=spec["nc"])
    return f, osp


def parameter_aligner_slice(po, spec, osp):
    """the finder's oracle slice completed with the aligner's part (robustifier, min_num_correspondences, sensor offset), in place"""
    osp.robustifier = po.ROBUST_CAUCHY if spec["cauchy"] else po.ROBUST_NONE; osp.chi_threshold = spec["tau"]; osp.min_num_correspondences = spec["mc"]
    osp.sensor_in_robot = (po.C.c_float * 3)(*spec["S"])
    return osp


def parameter_slice_processor(spec, f):
    rob = api.RobustifierCauchy(spec["tau"]) if spec["cauchy"] else None
    if any(spec["S"]):
        return api.AlignerSliceProcessorLaser2DWithSensor(f, sensor_in_robot=spec["S"], robustifier=rob, min_num_correspondences=spec["mc"])
    return api.AlignerSliceProcessorLaser2D(f, robustifier=rob, min_num_correspondences=spec["mc"])


# ---- the envelope's sample of the reference's own arithmetic ----------------------------------------------------------------------------------------
def perturbed_starts(x0, n=64, seed=1):
    """n start poses moved by one or two float32 ulps per component: all 26 one-ulp patterns of {-1, 0, +1}^3 first, then two-ulp patterns in a seeded order.
    What another compiler's -- or another summation order's -- last bit does to an arithmetic is sampled by running that arithmetic from these starts
    (rounds 4-5 used four of the one-ulp patterns; round 6's replay of the soak's violators showed four to be too thin: tests/replay_violators.py)."""
    x0 = np.asarray(x0, np.float32)
    one = [s for s in itertools.product((-1, 0, 1), repeat=3) if any(s)]
    two = [s for s in itertools.product((-2, -1, 0, 1, 2), repeat=3) if max(abs(v) for v in s) == 2]
    rng = np.random.default_rng(seed); rng.shuffle(two)
    out = []
    for steps in (one + two)[:n]:
        x = x0.copy()
        for k, st in enumerate(steps):
            for _ in range(abs(st)):
                x[k] = np.nextafter(np.float32(x[k]), np.float32(math.copysign(np.inf, st)))      # (both float32: a float64 direction would step in double and round back)
        assert x.dtype == np.float32 and not np.array_equal(x, x0)
        out.append((steps, x))
    return out


# Alignments of the eighteen-seed soak (420 trials per seed and test) whose DEFAULT-order result (tree sums) lies outside the envelope although the reference's own
# arithmetic, sampled at 64 perturbed starts, stays inside 1e-4 of the fp64 oracle: the tree order alone lands a pair on the other side of a gate
# (profiles/r06/violators_replay_r06.txt).  With "sum_order" 1 both equal the sequential fp32 oracle bit for bit, like every other alignment.  Named, not counted:
# (test, seed, trial, alignment) -> largest tolerated |device - fp64| in metres / radians
KNOWN_TREE_ORDER_DEVIATIONS = {
    ("structure", 17, 346, 0): 1.2e-4,         # 1.08e-4 m; 15 of 64 perturbed tree-order runs leave 1e-4, none of the reference's
    ("structure", 8675309, 66, 1): 3.0e-4,     # 2.80e-4 m; 64 of 64 tree-order runs, none of the reference's
}
# ... and the alignments where the reference arithmetic itself has no answer: its own runs from the 64 perturbed starts (or its two evaluations from the logged start)
# end more than 1e-2 m / 1e-2 rad from the fp64 oracle -- metres, for the first two.  Listed by the eighteen-seed soak (profiles/r06/fuzz_soak_18_seeds_r06i.log);
# with "sum_order" 1 the device still equals the sequential oracle bit for bit in each of them.
KNOWN_ILL_CONDITIONED = {
    ("parameters", 4711, 219, 0),      # one ulp on the start pose moves the sequential oracle by 4 m, the reference arithmetic by 14 m, the tree order by 30 m
    ("parameters", 7, 25, 0),          # the two fp32 evaluations themselves are 2.2 m from fp64; perturbed runs 17.8 m
    ("parameters", 42, 30, 0),         # perturbed sequential runs 3.1e-2 m / 3.0e-3 rad, the device 3.1e-2 m: the same spread
    # ... and seven whose two reference-arithmetic evaluations (sequential fp32, libm / no FMA) are THEMSELVES more than 1e-2 from the fp64 oracle -- runaway
    # iterations that round 5 passed as "within 3 x the reference arithmetic's own distance" (the device's distance in brackets)
    ("parameters", 1, 219, 0),         # oracles 16 m from fp64 (device 11 m)
    ("parameters", 5, 80, 0),          # 0.18 m / 2.3e-2 rad (0.12 m)
    ("parameters", 5, 415, 0),         # 5.9 m (0.75 m)
    ("parameters", 11, 255, 0),        # 1.4 km (1.6 km)
    ("parameters", 42, 415, 0),        # 0.87 m (2.0 m)
    ("parameters", 123, 204, 0),       # 1.9e-2 m (the same 1.9e-2 m)
    ("parameters", 8675309, 175, 0),   # 5.6 m (7.6 m)
    # ... and seven more of the same kind from eighteen FRESH seeds (profiles/r06/fuzz_soak_fresh_seeds_r06I.log; no alignment of those seeds lies outside the envelope)
    ("parameters", 303, 310, 0),       # oracles 4.4 m / 0.22 rad from fp64 (device 2.4 m / 0.14 rad)
    ("parameters", 505, 12, 0),        # 11.2 m (12.0 m)
    ("parameters", 606, 405, 0),       # 214 m (118 m)
    ("parameters", 808, 219, 0),       # 0.49 m (2.8e-2 m)
    ("parameters", 1234, 142, 0),      # 2.0 m (2.9 m)
    ("parameters", 4321, 159, 0),      # 1.3 m (0.32 m)
    ("parameters", 9999, 243, 0),      # 2.1 m (0.51 m)
}
