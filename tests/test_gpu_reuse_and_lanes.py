"""GPU tests of what the library KEEPS between calls (a batch's input block, its placement, a lane's events) and of the lanes of batches in flight:
everything kept must be invisible in the results.  Round 6: the cases the round-5 advisor named."""
import numpy as np
import pytest

from srrg2_laser_slam_2d_amd import api

pytestmark = pytest.mark.gpu


def _aligner(ctx, cols=1081, its=20):
    proj = api.PointNormal2fProjectorPolar(cols, -np.pi, np.pi, 0.3, 30.0)
    al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj), min_num_correspondences=10))
    return al


def test_same_batch_again_with_more_outputs_is_uploaded_again(small_workload):
    """Advisor (high, round 5): the kept input block lives in the lane's device scratch.  The same batch run again WITH statistics needs a bigger scratch; the
    reallocation holds nobody's inputs, and the upload must not be skipped (it was: the shadow compared equal).  Fresh context, so the scratch really grows."""
    wl = small_workload
    n = 1000
    fi = (np.arange(n, dtype=np.int32) % len(wl.x0)).reshape(1, n)
    x0 = wl.x0[fi[0]].astype(np.float32).copy()
    c = api.Context(0)
    try:
        al = _aligner(c)
        fixed = api.CloudSet(c, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c, wl.map_points)
        plain = al.compute_batch([fixed], [moving], x0, fixed_index=fi)                          # no statistics: a small scratch, the input block kept
        again = al.compute_batch([fixed], [moving], x0, fixed_index=fi)                          # the same inputs: the kept block is used
        with_stats = al.compute_batch([fixed], [moving], x0, fixed_index=fi, want_stats=True)    # + 28 B x 20 x n of statistics: the scratch grows
        assert np.array_equal(plain.pose, again.pose) and np.array_equal(plain.status, again.status)
        assert np.array_equal(plain.pose, with_stats.pose) and np.array_equal(plain.information, with_stats.information) and np.array_equal(plain.status, with_stats.status)
    finally:
        c.close()
    c2 = api.Context(0)
    try:
        al = _aligner(c2)
        fixed = api.CloudSet(c2, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c2, wl.map_points)
        fresh = al.compute_batch([fixed], [moving], x0, fixed_index=fi, want_stats=True)
        assert np.array_equal(fresh.pose, with_stats.pose) and np.array_equal(fresh.stats, with_stats.stats)
    finally:
        c2.close()


def test_asynchronous_zero_copy_batch_waits_for_its_own_lane(small_workload):
    """Advisor (medium, round 5): a small batch begun asynchronously writes its results to pinned memory and is retired by polling; when the launch outlives
    the 20 ms spin budget the fallback must wait for the LANE's stream (an event behind the launch), not for the context's idle one.  60 alignments of 10 000
    iterations against the 20k map take longer than that."""
    wl = small_workload
    n = 60
    fi = (np.arange(n, dtype=np.int32) % len(wl.x0)).reshape(1, n)
    x0 = wl.x0[fi[0]].astype(np.float32).copy()
    c = api.Context(0)
    try:
        c.set_option("align_path", 1)      # one workgroup per alignment: long-lived launch
        al = _aligner(c, its=10000)
        fixed = api.CloudSet(c, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c, wl.map_points)
        want = al.compute_batch([fixed], [moving], x0, fixed_index=fi)
        assert c.last_kernel_ms() > 25.0, "the launch must outlive the spin budget for this test to mean anything: %.1f ms" % c.last_kernel_ms()
        prep = al.prepare_batch([fixed], [moving], x0, fixed_index=fi)
        for _ in range(2):
            prep.begin()
            got = prep.wait()
            assert np.array_equal(got.status, want.status) and np.array_equal(got.pose, want.pose)
    finally:
        c.close()


def test_last_kernel_ms_follows_the_latest_timed_launch_after_lanes_swapped(small_workload):
    """Advisor (low, round 5): after an asynchronous begin the context works on the other lane; a timed finder call records that lane's events, and
    lsm2d_last_kernel_ms must read those -- not the waited batch's."""
    wl = small_workload
    c = api.Context(0)
    try:
        al = _aligner(c)
        fixed = api.CloudSet(c, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c, wl.map_points)
        prep = al.prepare_batch([fixed], [moving], wl.x0)
        prep.begin(); prep.wait()
        batch_ms = c.last_kernel_ms()
        finder = api.CorrespondenceFinderProjective2f(c, api.PointNormal2fProjectorPolar(1081, -np.pi, np.pi, 0.3, 30.0))
        finder.setFixed(fixed, 0); finder.setMoving(moving); finder.setLocalMapInSensor(wl.x0[0]); finder.compute()
        find_ms = c.last_kernel_ms()
        assert find_ms > 0.0 and find_ms < 0.5 * batch_ms, (find_ms, batch_ms)      # one finder pass against twenty iterations of six alignments
    finally:
        c.close()


def test_packed_batch_kept_placement_and_asynchronous_begin(small_workload):
    """Round 6 (late): a PACKED batch (1030 alignments in one dispatch round of 1024 workgroups, the lightest ones two to a workgroup) keeps both halves of its
    placement -- first and second alignment per workgroup -- for the same batch coming again (no estimate launch), makes them afresh for new start poses, and runs
    the same way when begun asynchronously, alone or beside another batch in flight (which gets no placement: the plain launch): every result the same bits."""
    wl = small_workload
    c = api.Context(0)
    try:
        al = _aligner(c, its=8)
        n = 1030
        fi = (np.arange(n, dtype=np.int32) % 6).reshape(1, n)
        x0 = wl.x0[fi[0]].astype(np.float32).copy(); x0[:, 0] += np.linspace(-0.02, 0.02, n, dtype=np.float32)
        fixed = api.CloudSet(c, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(c, wl.map_points)
        c.set_option("align_width", 512)
        ref = al.compute_batch([fixed], [moving], x0, fixed_index=fi)
        c.set_option("align_width", 0)
        prep = al.prepare_batch([fixed], [moving], x0, fixed_index=fi)
        runs = []
        for k in range(3):      # first: estimate by the round-3 assumption; second: by the first one's notes; third: kept
            runs.append(prep.run()); assert c.get_option("last_align_width") == 1024
            assert c.get_option("last_cull_estimate") == (1 if k < 2 else 0), k
        x1 = x0.copy(); x1[:, 1] += np.float32(0.01)
        prep.set_init_poses(x1); moved = prep.run()
        assert c.get_option("last_align_width") == 1024 and c.get_option("last_cull_estimate") == 1
        prep.set_init_poses(x0); prep.begin(); runs.append(prep.wait(copy=True))
        assert c.get_option("last_align_width") == 1024
        prep2 = al.prepare_batch([fixed], [moving], x0, fixed_index=fi)
        prep.begin(); prep2.begin(); runs.append(prep.wait(copy=True)); runs.append(prep2.wait(copy=True))
        for r in runs:
            assert np.array_equal(r.pose, ref.pose) and np.array_equal(r.information, ref.information) and np.array_equal(r.status, ref.status) and np.array_equal(r.iterations, ref.iterations)
        assert np.all(ref.status == 0) and np.all(moved.status == 0) and np.abs(moved.pose - ref.pose).max() < 1e-4      # (noise-free data: the same answer from the other start, to the bar)
    finally:
        c.close()
