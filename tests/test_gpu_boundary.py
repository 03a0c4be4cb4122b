"""GPU parity tests, row b: the C ABI's error paths and limits, memory, ownership, the SRRG adapters and the C++ hosts compiled and run on the GPU.

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
