"""Synthetic worlds, local maps and laser scans for the scan-matching hot path.

The reference ships only a toy generator (circle + corner, glibc ``rand()``:
``srrg2_laser_slam_2d/apps/synthetic_scene_generator.cpp:36-54,167-181``); SURVEY.md section 8(d) fixes the
benchmark inputs used here: a closed 40 m x 30 m room with 12 axis-aligned pillars, a local map
of N points at uniform arc-length spacing (segment-major order), 1081-beam scans (-135..+135 deg,
0.25 deg step, 0.1..30 m) ray-cast from random free-space poses, initial guesses perturbed by
U(-0.05, 0.05) in x, y [m] and theta [rad] (the scale of synthetic_scene_generator.cpp:167-178).

Everything is generated from an explicit 64-bit seed by a counter-based splitmix64 stream written
here (never ``rand()``), so fixtures are reproducible across platforms and numpy versions.
All clouds are float32 ``[N, 4]`` rows ``(x, y, nx, ny)`` -- the PointNormal2f payload.
"""
from __future__ import annotations

import dataclasses

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


class Stream:
    """Counter-based uniform stream: value i = splitmix64(splitmix64(seed) + i)."""

    def __init__(self, seed: int, salt: int = 0):
        s = np.array([(int(seed) * 0x632BE59BD9B4E019 + int(salt) * 0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF],
                     dtype=np.uint64)
        self._base = _splitmix64(s)[0]
        self._ctr = 0

    def uniform(self, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        idx = np.arange(self._ctr, self._ctr + n, dtype=np.uint64)
        self._ctr += n
        with np.errstate(over="ignore"):
            bits = _splitmix64((self._base + idx) & _M64)
        u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
        return lo + (hi - lo) * u

    def normal(self, n: int) -> np.ndarray:
        u1 = self.uniform(n)
        u2 = self.uniform(n)
        return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


@dataclasses.dataclass
class World:
    """Line segments ``a -> b`` with unit normals pointing into free space."""
    a: np.ndarray        # [S, 2] float64
    b: np.ndarray        # [S, 2]
    n: np.ndarray        # [S, 2]
    pillars: np.ndarray  # [P, 4] (xmin, ymin, xmax, ymax)
    half_extent: tuple = (20.0, 15.0)

    @property
    def lengths(self) -> np.ndarray:
        return np.linalg.norm(self.b - self.a, axis=1)


def make_world(seed: int = 0, n_pillars: int = 12) -> World:
    hx, hy = 20.0, 15.0
    st = Stream(seed, salt=1)
    pillars = []
    guard = 0
    while len(pillars) < n_pillars and guard < 10000:
        guard += 1
        w, h = st.uniform(2, 0.5, 3.0)
        cx = st.uniform(1, -hx + 2.5, hx - 2.5)[0]
        cy = st.uniform(1, -hy + 2.5, hy - 2.5)[0]
        r = (cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2)
        if r[0] - 2.0 < 0.0 < r[2] + 2.0 and r[1] - 2.0 < 0.0 < r[3] + 2.0:
            continue  # within 2 m of the origin
        if any(not (r[2] + 0.6 < q[0] or q[2] + 0.6 < r[0] or r[3] + 0.6 < q[1] or q[3] + 0.6 < r[1]) for q in pillars):
            continue  # keep pillars disjoint
        pillars.append(r)
    A, B, N = [], [], []

    def seg(ax, ay, bx, by, nx, ny):
        A.append((ax, ay)); B.append((bx, by)); N.append((nx, ny))

    # room walls, normals inward
    seg(-hx, -hy, hx, -hy, 0, 1)
    seg(hx, -hy, hx, hy, -1, 0)
    seg(hx, hy, -hx, hy, 0, -1)
    seg(-hx, hy, -hx, -hy, 1, 0)
    for (x0, y0, x1, y1) in pillars:  # pillar faces, normals outward
        seg(x0, y0, x1, y0, 0, -1)
        seg(x1, y0, x1, y1, 1, 0)
        seg(x1, y1, x0, y1, 0, 1)
        seg(x0, y1, x0, y0, -1, 0)
    return World(np.array(A, float), np.array(B, float), np.array(N, float), np.array(pillars, float), (hx, hy))


def make_circle_corner_world_points() -> np.ndarray:
    """The reference's own toy scene (synthetic_scene_generator.cpp:36-54, 240-283): circle r=3.5 m
    with 2048 points + 2 m x 3 m corner with 1024 points moved to (2, 0, pi/4).  The reference leaves
    the normals zero and recomputes them downstream; here they are analytic (circle: inward; corner
    legs: the side facing the circle centre)."""
    n = 2048
    ang = np.arange(n) * (2 * np.pi / n)
    circle = np.stack([3.5 * np.cos(ang), 3.5 * np.sin(ang), -np.cos(ang), -np.sin(ang)], 1)
    npts = 1024
    step = 5.0 / npts
    n0 = int(2.0 / step)
    n1 = npts - n0
    l0 = np.stack([step * np.arange(n0), np.zeros(n0), np.zeros(n0), np.ones(n0)], 1)
    l1 = np.stack([np.zeros(n1 - 1), -step * np.arange(1, n1), -np.ones(n1 - 1), np.zeros(n1 - 1)], 1)
    corner = np.concatenate([l0, l1], 0)
    c, s = np.cos(np.pi / 4), np.sin(np.pi / 4)
    R = np.array([[c, -s], [s, c]])
    corner[:, :2] = corner[:, :2] @ R.T + np.array([2.0, 0.0])
    corner[:, 2:] = corner[:, 2:] @ R.T
    # orient corner normals towards the origin side they are seen from
    flip = np.sum(corner[:, 2:] * (-corner[:, :2]), 1) < 0
    corner[flip, 2:] *= -1
    return np.concatenate([circle, corner], 0).astype(np.float32)


def make_map(world: World, n_points: int, noise_sigma: float = 0.0, seed: int = 0, shuffle: bool = False) -> np.ndarray:
    """N points at uniform arc-length spacing over all segments, segment-major order."""
    L = world.lengths
    cum = np.concatenate([[0.0], np.cumsum(L)])
    s = (np.arange(n_points) + 0.5) * (cum[-1] / n_points)
    k = np.clip(np.searchsorted(cum, s, side="right") - 1, 0, len(L) - 1)
    t = (s - cum[k]) / L[k]
    p = world.a[k] + (world.b[k] - world.a[k]) * t[:, None]
    if noise_sigma > 0:
        st = Stream(seed, salt=2)
        p = p + noise_sigma * np.stack([st.normal(n_points), st.normal(n_points)], 1)
    out = np.concatenate([p, world.n[k]], 1).astype(np.float32)
    if shuffle:
        st = Stream(seed, salt=3)
        out = out[np.argsort(st.uniform(n_points), kind="stable")]
    return np.ascontiguousarray(out)


def _free(world: World, xy: np.ndarray, clearance: float) -> np.ndarray:
    hx, hy = world.half_extent
    ok = (np.abs(xy[:, 0]) < hx - clearance) & (np.abs(xy[:, 1]) < hy - clearance)
    for (x0, y0, x1, y1) in world.pillars:
        ok &= ~((xy[:, 0] > x0 - clearance) & (xy[:, 0] < x1 + clearance) &
                (xy[:, 1] > y0 - clearance) & (xy[:, 1] < y1 + clearance))
    return ok


def sample_poses(world: World, n: int, seed: int = 0, clearance: float = 0.6) -> np.ndarray:
    """Sensor poses (x, y, theta) uniform over free space, heading uniform.  float64 [n, 3]."""
    st = Stream(seed, salt=4)
    hx, hy = world.half_extent
    out = np.zeros((0, 3))
    while len(out) < n:
        m = max(2 * (n - len(out)), 16)
        c = np.stack([st.uniform(m, -hx, hx), st.uniform(m, -hy, hy), st.uniform(m, -np.pi, np.pi)], 1)
        out = np.concatenate([out, c[_free(world, c, clearance)]], 0)
    return out[:n]


def scan_angles(n_beams: int = 1081, fov_deg: float = 270.0) -> np.ndarray:
    return np.deg2rad(-fov_deg / 2 + np.arange(n_beams) * (fov_deg / (n_beams - 1)))


def make_scan_ranges(world: World, poses: np.ndarray, n_beams: int = 1081, angle_min: float = -0.75 * np.pi,
                     angle_max: float = 0.75 * np.pi, noise_sigma: float = 0.0, seed: int = 0, chunk: int = 64) -> np.ndarray:
    """Raw LaserMessage-style ranges float32 [n, n_beams] for the preprocessor: beam c looks along
    (c - n_beams/2) * (angle_max - angle_min) / n_beams (the bearing convention of the reference's sensor matrix,
    sensor_processing/raw_data_preprocessor_projective_2d.cpp:87-90); no hit -> +inf."""
    res = (angle_max - angle_min) / n_beams
    ang = (np.arange(n_beams) - n_beams / 2.0) * res
    a, d = world.a, world.b - world.a
    out = np.empty((len(poses), n_beams), np.float32)
    st = Stream(seed, salt=12)
    for lo in range(0, len(poses), chunk):
        P = poses[lo:lo + chunk]
        th = P[:, 2:3] + ang[None, :]
        dx, dy = np.cos(th), np.sin(th)
        ox, oy = P[:, 0, None, None], P[:, 1, None, None]
        den = dx[..., None] * d[None, None, :, 1] - dy[..., None] * d[None, None, :, 0]
        ex, ey = a[None, None, :, 0] - ox, a[None, None, :, 1] - oy
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (ex * d[None, None, :, 1] - ey * d[None, None, :, 0]) / den
            u = (ex * dy[..., None] - ey * dx[..., None]) / den
        hit = (np.abs(den) > 1e-12) & (t > 0) & (u >= 0) & (u <= 1)
        r = np.min(np.where(hit, t, np.inf), axis=2)
        if noise_sigma > 0:
            r = r + noise_sigma * st.normal(r.size).reshape(r.shape)
        out[lo:lo + chunk] = r
    return out


def make_scans(world: World, poses: np.ndarray, n_beams: int = 1081, fov_deg: float = 270.0,
               range_min: float = 0.1, range_max: float = 30.0, noise_sigma: float = 0.0, seed: int = 0,
               chunk: int = 64):
    """Ray-cast ``len(poses)`` scans.  Returns (points float32 [sum, 4] in the SENSOR frame with
    analytic normals, offsets int32 [n+1]); beams without a hit inside the range limits are dropped."""
    ang = scan_angles(n_beams, fov_deg)
    a, d, nrm = world.a, world.b - world.a, world.n
    pts, counts = [], []
    st = Stream(seed, salt=5)
    for lo in range(0, len(poses), chunk):
        P = poses[lo:lo + chunk]
        th = P[:, 2:3] + ang[None, :]                       # [c, B]
        dx, dy = np.cos(th), np.sin(th)
        ox, oy = P[:, 0, None, None], P[:, 1, None, None]     # [c,1,1]
        # ray o + t*dir hits segment a + u*d:  t = cross(a-o, d)/cross(dir, d), u = cross(a-o, dir)/cross(dir, d)
        den = dx[..., None] * d[None, None, :, 1] - dy[..., None] * d[None, None, :, 0]   # [c,B,S]
        ex, ey = a[None, None, :, 0] - ox, a[None, None, :, 1] - oy
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (ex * d[None, None, :, 1] - ey * d[None, None, :, 0]) / den
            u = (ex * dy[..., None] - ey * dx[..., None]) / den
        hit = (np.abs(den) > 1e-12) & (t > 0) & (u >= 0) & (u <= 1)
        t = np.where(hit, t, np.inf)
        k = np.argmin(t, axis=2)                               # [c,B]
        r = np.take_along_axis(t, k[..., None], 2)[..., 0]
        if noise_sigma > 0:
            r = r + noise_sigma * st.normal(r.size).reshape(r.shape)
        valid = np.isfinite(r) & (r >= range_min) & (r <= range_max)
        # sensor-frame points and normals (world normal rotated by R^T)
        px, py = r * np.cos(ang)[None, :], r * np.sin(ang)[None, :]
        nw = nrm[k]                                            # [c,B,2]
        c_, s_ = np.cos(P[:, 2])[:, None], np.sin(P[:, 2])[:, None]
        nx = c_ * nw[..., 0] + s_ * nw[..., 1]
        ny = -s_ * nw[..., 0] + c_ * nw[..., 1]
        for i in range(len(P)):
            v = valid[i]
            pts.append(np.stack([px[i, v], py[i, v], nx[i, v], ny[i, v]], 1).astype(np.float32))
            counts.append(int(v.sum()))
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    points = np.concatenate(pts, 0) if pts else np.zeros((0, 4), np.float32)
    return np.ascontiguousarray(points), offsets


def v2t(v):
    c, s = np.cos(v[2]), np.sin(v[2])
    return np.array([[c, -s, v[0]], [s, c, v[1]], [0, 0, 1.0]])


def t2v(T):
    return np.array([T[0, 2], T[1, 2], np.arctan2(T[1, 0], T[0, 0])])


def invert_poses(p: np.ndarray) -> np.ndarray:
    """(x, y, theta) of the inverse isometries, float64 [n, 3]."""
    c, s = np.cos(p[:, 2]), np.sin(p[:, 2])
    return np.stack([-(c * p[:, 0] + s * p[:, 1]), -(-s * p[:, 0] + c * p[:, 1]), -p[:, 2]], 1)


def compose_poses(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    c, s = np.cos(a[:, 2]), np.sin(a[:, 2])
    th = a[:, 2] + b[:, 2]
    th = (th + np.pi) % (2 * np.pi) - np.pi
    return np.stack([a[:, 0] + c * b[:, 0] - s * b[:, 1], a[:, 1] + s * b[:, 0] + c * b[:, 1], th], 1)


def initial_guesses(true_sensor_poses: np.ndarray, seed: int = 0, scale: float = 0.05):
    """Returns (x_true, x0): map-in-sensor poses (the aligner estimate ``moving_in_fixed`` with
    fixed = scan, moving = map) for the true pose and for T0 = T* . v2t(delta), delta ~ U(-scale, scale)^3."""
    st = Stream(seed, salt=6)
    n = len(true_sensor_poses)
    delta = st.uniform(3 * n, -scale, scale).reshape(n, 3)
    t0 = compose_poses(true_sensor_poses, delta)
    return invert_poses(true_sensor_poses), invert_poses(t0)


@dataclasses.dataclass
class Workload:
    world: World
    map_points: np.ndarray      # float32 [N_m, 4]   moving cloud (shared local map)
    scan_points: np.ndarray     # float32 [sum, 4]   fixed clouds, packed
    scan_offsets: np.ndarray    # int32 [n+1]
    x_true: np.ndarray          # float64 [n, 3]
    x0: np.ndarray              # float32 [n, 3]


def make_workload(n_scans: int, n_map: int, seed: int = 0, n_beams: int = 1081, map_noise: float = 0.0,
                  scan_noise: float = 0.0, pose_seed_offset: int = 0, world: World | None = None,
                  map_points: np.ndarray | None = None) -> Workload:
    world = world if world is not None else make_world(seed)
    if map_points is None:
        map_points = make_map(world, n_map, noise_sigma=map_noise, seed=seed)
    poses = sample_poses(world, n_scans, seed=seed + 7919 * pose_seed_offset)
    scans, offs = make_scans(world, poses, n_beams=n_beams, noise_sigma=scan_noise, seed=seed + pose_seed_offset)
    x_true, x0 = initial_guesses(poses, seed=seed + pose_seed_offset)
    return Workload(world, map_points, scans, offs, x_true, x0.astype(np.float32))
