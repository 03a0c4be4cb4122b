"""ctypes view of the CPU ORACLE (oracle/lsm2d_oracle.h) -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the oracle is a restatement of the reference algorithm (see lsm2d_oracle.h for
the file:line map); no reference golden vector exists for this path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liblsm2d_oracle.so")

FINDER_PROJECTIVE, FINDER_NN, FINDER_DISTMAP, FINDER_KDTREE_APPROX = 0, 1, 2, 3
ROBUST_NONE, ROBUST_CAUCHY = 0, 1
SUCCESS, NOT_ENOUGH_CORRESPONDENCES, NOT_ENOUGH_INLIERS, SINGULAR_H, BAD_ARGUMENT = 0, 1, 2, 3, -1


class Projector(C.Structure):
    _fields_ = [("canvas_cols", C.c_int), ("angle_min", C.c_float), ("angle_max", C.c_float),
                ("range_min", C.c_float), ("range_max", C.c_float), ("col_offset", C.c_float)]


class SliceParams(C.Structure):
    _fields_ = [("finder", C.c_int), ("projector", Projector), ("point_distance", C.c_float),
                ("normal_cos", C.c_float), ("max_distance", C.c_float), ("resolution", C.c_float),
                ("robustifier", C.c_int), ("chi_threshold", C.c_float),
                ("min_num_correspondences", C.c_int), ("sensor_in_robot", C.c_float * 3),
                ("kd_max_leaf_range", C.c_float), ("kd_min_leaf_points", C.c_int)]


class AlignerParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("min_num_inliers", C.c_int), ("damping", C.c_float),
                ("has_prior", C.c_int), ("prior_z", C.c_float * 3), ("prior_omega", C.c_float * 9), ("device_order", C.c_int),
                ("termination_chi_epsilon", C.c_float), ("enable_inlier_only_runs", C.c_int), ("keep_only_inlier_correspondences", C.c_int)]


class IterStats(C.Structure):
    _fields_ = [("n_corr", C.c_int), ("n_in", C.c_int), ("n_out", C.c_int),
                ("chi_in", C.c_float), ("chi_out", C.c_float), ("pair_digest_lo", C.c_uint), ("pair_digest_hi", C.c_uint)]

    @property
    def pair_digest(self) -> int:
        """order-independent 64-bit digest of the iteration's correspondence set (lsm2d_oracle.h)"""
        return (int(self.pair_digest_hi) << 32) | int(self.pair_digest_lo)


def _host_stamp() -> str:
    """-march=native ties the binary to this host's CPU: rebuild when the tree lands on another box."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("flags"):
                return hashlib.sha1(line.encode()).hexdigest()
    except OSError:
        pass
    return "unknown"


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    srcs = [os.path.join(_HERE, f) for f in ("lsm2d_oracle.c", "lsm2d_oracle_impl.inc", "lsm2d_oracle.h")]
    stamp_path = os.path.join(_HERE, "_build", "host.stamp")
    stamp = _host_stamp()
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if not stale:
        try:
            stale = open(stamp_path).read().strip() != stamp
        except OSError:
            stale = True
    if stale:
        subprocess.run(["make", "-B", "-C", _HERE], check=True, capture_output=True)
        with open(stamp_path, "w") as fh:
            fh.write(stamp)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("LSM2D_ORACLE_LIB")      # e.g. the -fsanitize build from `make -C oracle asan`
        _lib = C.CDLL(override if override else build())
        _lib.lsmo_atan2f.restype = C.c_float
        _lib.lsmo_atan2f.argtypes = [C.c_float, C.c_float]
        _lib.lsmo_logf_fixed.restype = C.c_float
        _lib.lsmo_logf_fixed.argtypes = [C.c_float]
        _lib.lsmo_sincosf.restype = None
        _lib.lsmo_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib.lsmo_pair_hash.restype = C.c_ulonglong
        _lib.lsmo_pair_hash.argtypes = [C.c_uint, C.c_uint, C.c_uint]
    return _lib


def pair_digest(pairs, slice_index: int = 0) -> int:
    """wrapping 64-bit sum of lsmo_pair_hash(slice, fixed_idx, moving_idx) over int32 pairs [k, 2] (numpy restatement of the hash; the C
    definition is checked against it in tests/test_oracle.py)"""
    p = np.ascontiguousarray(pairs, np.int64).reshape(-1, 2)
    M = np.uint32
    with np.errstate(over="ignore"):
        f = p[:, 0].astype(np.uint32); m = p[:, 1].astype(np.uint32)
        rot = lambda v, r: (v << M(r)) | (v >> M(32 - r))
        a = f * M(0x9E3779B1); b = (m ^ M((slice_index * 0x632BE5AB) & 0xFFFFFFFF)) * M(0x85EBCA77)
        lo = a ^ rot(b, 13); hi = b ^ rot(a, 19)
        lo = lo + (rot(lo, 17) ^ b); hi = hi + (rot(hi, 11) ^ a)
        tot = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
        return int(tot.sum(dtype=np.uint64)) & 0xFFFFFFFFFFFFFFFF


def log_fixed(x):
    """float32 array through the oracle's fixed-sequence lsmo_logf_fixed."""
    x = np.atleast_1d(np.asarray(x, np.float32)); L = lib()
    return np.array([L.lsmo_logf_fixed(float(v)) for v in x], np.float32)


def sincos(x):
    """(sin, cos) float32 arrays through the oracle's fixed-sequence lsmo_sincosf."""
    x = np.atleast_1d(np.asarray(x, np.float32)); L = lib()
    s = np.empty_like(x); c = np.empty_like(x); a, b = C.c_float(), C.c_float()
    for i, v in enumerate(x):
        L.lsmo_sincosf(float(v), C.byref(a), C.byref(b)); s[i], c[i] = a.value, b.value
    return s, c


def slice_params(finder=FINDER_PROJECTIVE, canvas_cols=1081, angle_min=-np.pi, angle_max=np.pi, range_min=0.3,
                 range_max=30.0, col_offset=0.0, point_distance=0.5, normal_cos=0.8, max_distance=0.5,
                 resolution=0.05, robustifier=ROBUST_NONE, chi_threshold=0.05, min_num_correspondences=10,
                 sensor_in_robot=(0.0, 0.0, 0.0), kd_max_leaf_range=1e-2, kd_min_leaf_points=20) -> SliceParams:
    sp = SliceParams()
    sp.finder = finder
    sp.projector = Projector(canvas_cols, angle_min, angle_max, range_min, range_max, col_offset)
    sp.point_distance, sp.normal_cos, sp.max_distance, sp.resolution = point_distance, normal_cos, max_distance, resolution
    sp.robustifier, sp.chi_threshold, sp.min_num_correspondences = robustifier, chi_threshold, min_num_correspondences
    sp.sensor_in_robot = (C.c_float * 3)(*sensor_in_robot)
    sp.kd_max_leaf_range, sp.kd_min_leaf_points = kd_max_leaf_range, kd_min_leaf_points
    return sp


def aligner_params(max_iterations=20, min_num_inliers=10, damping=0.0, prior_z=None, prior_omega=None, device_order=False,
                   termination_chi_epsilon=0.0, enable_inlier_only_runs=False, keep_only_inlier_correspondences=False) -> AlignerParams:
    """device_order: sum H, b and the statistics in the HIP kernels' order (fp32 mirror only) -> bit-identical to the device."""
    ap = AlignerParams()
    ap.device_order = 1 if device_order else 0
    ap.termination_chi_epsilon = termination_chi_epsilon
    ap.enable_inlier_only_runs = 1 if enable_inlier_only_runs else 0
    ap.keep_only_inlier_correspondences = 1 if keep_only_inlier_correspondences else 0
    ap.max_iterations, ap.min_num_inliers, ap.damping = max_iterations, min_num_inliers, damping
    ap.has_prior = 0 if prior_z is None else 1
    if prior_z is not None:
        ap.prior_z = (C.c_float * 3)(*prior_z)
        ap.prior_omega = (C.c_float * 9)(*np.asarray(prior_omega, np.float32).ravel())
    return ap


def _pts(a):
    a = np.ascontiguousarray(a, np.float32)
    assert a.ndim == 2 and a.shape[1] == 4
    return a, a.ctypes.data_as(C.c_void_p)


def _real(double):
    """double: False -> fp32 mirror `_f`; True -> fp64 truth `_d`; "ref" -> fp32 in the reference's own arithmetic `_r`
    (libm atan2f / sinf / cosf / logf, no FMA, Eigen's association; lsm2d_oracle.h)."""
    if isinstance(double, str):
        assert double == "ref", double
        return np.float32, "_r"
    return (np.float64, "_d") if double else (np.float32, "_f")


def atan2f(y, x):
    y = np.asarray(y, np.float32).ravel(); x = np.asarray(x, np.float32).ravel()
    L = lib()
    return np.array([L.lsmo_atan2f(float(a), float(b)) for a, b in zip(y, x)], np.float32)


def project(pr: Projector, cloud, pose, double=False):
    dt, sfx = _real(double)
    cloud, pc = _pts(cloud)
    cols = pr.canvas_cols
    src = np.empty(cols, np.int32); depth = np.empty(cols, dt); xyn = np.empty((cols, 4), dt)
    pose = np.ascontiguousarray(pose, dt)
    rc = getattr(lib(), "lsmo_project" + sfx)(C.byref(pr), pc, len(cloud), pose.ctypes.data_as(C.c_void_p),
                                              src.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                                              xyn.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return src, depth, xyn


def find(sp: SliceParams, fixed, moving, pose, double=False, brute=False):
    """Correspondences int32 [k, 2] = (fixed_idx, moving_idx)."""
    dt, sfx = _real(double)
    fixed, pf = _pts(fixed); moving, pm = _pts(moving)
    cap = sp.projector.canvas_cols if sp.finder == FINDER_PROJECTIVE else len(moving)
    out = np.empty((max(cap, 1), 2), np.int32)
    pose = np.ascontiguousarray(pose, dt)
    name = {FINDER_PROJECTIVE: "lsmo_find_projective", FINDER_NN: "lsmo_find_nn_brute" if brute else "lsmo_find_nn",
            FINDER_DISTMAP: "lsmo_find_distmap", FINDER_KDTREE_APPROX: "lsmo_find_kdtree"}[sp.finder]
    k = getattr(lib(), name + sfx)(C.byref(sp), pf, len(fixed), pm, len(moving), pose.ctypes.data_as(C.c_void_p),
                                   out.ctypes.data_as(C.c_void_p))
    assert k >= 0, k
    return out[:k].copy()


def linearize(sp: SliceParams, fixed, moving, corr, pose, double=False):
    dt, sfx = _real(double)
    fixed, pf = _pts(fixed); moving, pm = _pts(moving)
    corr = np.ascontiguousarray(corr, np.int32)
    H = np.empty(9, dt); b = np.empty(3, dt); st = IterStats()
    pose = np.ascontiguousarray(pose, dt)
    rc = getattr(lib(), "lsmo_linearize" + sfx)(C.byref(sp), pf, pm, corr.ctypes.data_as(C.c_void_p), len(corr),
                                                pose.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p),
                                                b.ctypes.data_as(C.c_void_p), C.byref(st))
    assert rc == 0
    return H.reshape(3, 3), b, st


def linearize_device_order(sp: SliceParams, fixed, moving, corr, pose):
    """H, b, stats summed in the order lsm2d_linearize's launch forms them (workgroups of 256, one thread per pair up to 1024 groups)."""
    fixed, pf = _pts(fixed); moving, pm = _pts(moving)
    corr = np.ascontiguousarray(corr, np.int32)
    H = np.empty(9, np.float32); b = np.empty(3, np.float32); st = IterStats()
    pose = np.ascontiguousarray(pose, np.float32)
    blocks = min(1024, max(1, (len(corr) + 255) // 256))
    rc = lib().lsmo_linearize_device_order_f(C.byref(sp), pf, pm, corr.ctypes.data_as(C.c_void_p), None, len(corr), pose.ctypes.data_as(C.c_void_p),
                                             256 * blocks, 256, H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(st))
    assert rc == 0
    return H.reshape(3, 3), b, st


def error_jacobian(f, m, pose):
    f = np.ascontiguousarray(f, np.float32); m = np.ascontiguousarray(m, np.float32)
    e = np.empty(3); J = np.empty(9); pose = np.ascontiguousarray(pose, np.float64)
    lib().lsmo_error_jacobian_d(f.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p), pose.ctypes.data_as(C.c_void_p),
                                e.ctypes.data_as(C.c_void_p), J.ctypes.data_as(C.c_void_p))
    return e, J.reshape(3, 3)


def solve_update(H, b, pose, damping=0.0, double=False):
    dt, sfx = _real(double)
    H = np.ascontiguousarray(H, dt).ravel(); b = np.ascontiguousarray(b, dt); pose = np.array(pose, dt)
    dx = np.empty(3, dt)
    fn = getattr(lib(), "lsmo_solve_update" + sfx)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_double if double is True else C.c_float, C.c_void_p, C.c_void_p]
    rc = fn(H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), damping, pose.ctypes.data_as(C.c_void_p),
            dx.ctypes.data_as(C.c_void_p))
    return rc, pose, dx


def align(ap: AlignerParams, slices, fixed, moving, x0, double=False, want_pairs=False):
    """Multi-slice alignment. ``slices``: list of SliceParams; ``fixed``/``moving``: lists of clouds.
    Returns dict(status, pose, H, stats[list of IterStats], iterations); with want_pairs also pairs = per slice the int32 [k, 2] correspondences
    the aligner leaves behind (last iteration started; only its inliers under keep_only_inlier_correspondences)."""
    dt, sfx = _real(double)
    n = len(slices)
    sp = (SliceParams * n)(*slices)
    fx = [_pts(f) for f in fixed]; mv = [_pts(m) for m in moving]
    pf = (C.c_void_p * n)(*[p for _, p in fx]); pm = (C.c_void_p * n)(*[p for _, p in mv])
    nf = (C.c_int * n)(*[len(a) for a, _ in fx]); nm = (C.c_int * n)(*[len(a) for a, _ in mv])
    x0 = np.ascontiguousarray(x0, dt); xo = np.empty(3, dt); H = np.empty(9, dt)
    stats = (IterStats * (max(ap.max_iterations, 1) * (2 if ap.enable_inlier_only_runs else 1)))(); its = C.c_int(0)
    if not want_pairs:
        rc = getattr(lib(), "lsmo_align" + sfx)(C.byref(ap), n, sp, pf, nf, pm, nm, x0.ctypes.data_as(C.c_void_p),
                                                xo.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p), stats, C.byref(its))
        return dict(status=rc, pose=xo, H=H.reshape(3, 3), stats=[stats[i] for i in range(its.value)], iterations=its.value)
    bufs = [np.empty((max(s_.projector.canvas_cols if s_.finder == FINDER_PROJECTIVE else len(m_[0]), 1), 2), np.int32) for s_, m_ in zip(slices, mv)]
    pp = (C.c_void_p * n)(*[b.ctypes.data_as(C.c_void_p) for b in bufs]); npairs = (C.c_int * n)()
    rc = getattr(lib(), "lsmo_align_pairs" + sfx)(C.byref(ap), n, sp, pf, nf, pm, nm, x0.ctypes.data_as(C.c_void_p),
                                                  xo.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p), stats, C.byref(its), pp, npairs)
    return dict(status=rc, pose=xo, H=H.reshape(3, 3), stats=[stats[i] for i in range(its.value)], iterations=its.value,
                pairs=[bufs[s_][: npairs[s_]].copy() for s_ in range(n)])


def align_batch(ap: AlignerParams, sp: SliceParams, fixed_packed, fixed_offsets, moving, x0, n_threads=1, thread_times=None):
    """fp32 batch over a shared moving cloud (cpu_baseline 'port').  Returns (pose [n,3], H [n,3,3], status [n], last stats).
    thread_times: a dict that receives every worker's wall seconds and alignment count (the workers share one atomic work counter)."""
    fixed_packed, pf = _pts(fixed_packed); moving, pm = _pts(moving)
    offs = np.ascontiguousarray(fixed_offsets, np.int32); n = len(offs) - 1
    x0 = np.ascontiguousarray(x0, np.float32).reshape(n, 3)
    xo = np.empty((n, 3), np.float32); H = np.empty((n, 9), np.float32); status = np.empty(n, np.int32)
    last = (IterStats * max(n, 1))()
    nt = max(1, min(int(n_threads), max(n, 1)))
    secs = np.zeros(nt, np.float64); jobs = np.zeros(nt, np.int32)
    lib().lsmo_align_batch_timed_f(C.byref(ap), C.byref(sp), pf, offs.ctypes.data_as(C.c_void_p), n, pm, len(moving),
                                   x0.ctypes.data_as(C.c_void_p), xo.ctypes.data_as(C.c_void_p), H.ctypes.data_as(C.c_void_p),
                                   status.ctypes.data_as(C.c_void_p), last, int(n_threads), secs.ctypes.data_as(C.c_void_p), jobs.ctypes.data_as(C.c_void_p))
    if thread_times is not None:
        thread_times.update(seconds=secs, alignments=jobs)
    return xo, H.reshape(n, 3, 3), status, [last[i] for i in range(n)]


def clip_scene(pr: Projector, scene, robot_in_local_map, sensor_in_robot=(0.0, 0.0, 0.0), double=False):
    """SceneClipperProjective2D::compute. Returns (clipped float32 [k,4] in the robot frame, source indices int32 [k])."""
    dt, sfx = _real(double)
    scene, ps = _pts(scene)
    out = np.empty((pr.canvas_cols, 4), np.float32); src = np.empty(pr.canvas_cols, np.int32)
    r = np.ascontiguousarray(robot_in_local_map, dt); s_ = np.ascontiguousarray(sensor_in_robot, dt)
    k = getattr(lib(), "lsmo_clip_scene" + sfx)(C.byref(pr), ps, len(scene), r.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p),
                                               out.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p))
    assert k >= 0, k
    return out[:k].copy(), src[:k].copy()


def clip_scene_voxelized(pr: Projector, scene, robot_in_local_map, sensor_in_robot=(0.0, 0.0, 0.0), voxelize_resolution=0.05):
    """SceneClipperProjective2D::compute, voxelize_resolution > 0 branch (fp32 mirror).  Returns clipped float32 [k,4] in the robot frame."""
    scene, ps = _pts(scene)
    out = np.empty((pr.canvas_cols, 4), np.float32)
    r = np.ascontiguousarray(robot_in_local_map, np.float32); s_ = np.ascontiguousarray(sensor_in_robot, np.float32)
    fn = lib().lsmo_clip_scene_voxelized_f
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    k = fn(C.cast(C.byref(pr), C.c_void_p), ps, len(scene), r.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p), float(voxelize_resolution),
           out.ctypes.data_as(C.c_void_p))
    assert k >= 0, k
    return out[:k].copy()


def merge_scene(pr: Projector, scene, meas, measurement_in_scene, merge_threshold=0.2, double=False):
    """MergerProjective2D::compute. Returns (new scene float32 [n',4], counts (new, merged, replaced))."""
    dt, sfx = _real(double)
    scene = np.ascontiguousarray(scene, np.float32); meas, pm = _pts(meas)
    buf = np.zeros((len(scene) + pr.canvas_cols, 4), np.float32); buf[:len(scene)] = scene
    m = np.ascontiguousarray(measurement_in_scene, dt); counts = (C.c_int * 3)()
    fn = getattr(lib(), "lsmo_merge_scene" + sfx)
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_double if double is True else C.c_float, C.c_void_p]
    n = fn(C.cast(C.byref(pr), C.c_void_p), buf.ctypes.data_as(C.c_void_p), len(scene), pm, len(meas), m.ctypes.data_as(C.c_void_p), merge_threshold, counts)
    assert n >= 0, n
    return buf[:n].copy(), tuple(counts)


class Preprocessor(C.Structure):
    _fields_ = [("n_beams", C.c_int), ("angle_min", C.c_float), ("angle_max", C.c_float), ("range_min", C.c_float),
                ("range_max", C.c_float), ("normal_point_distance", C.c_float), ("normal_min_points", C.c_int),
                ("voxelize_resolution", C.c_float)]


def preprocess_scan(pp: Preprocessor, ranges) -> np.ndarray:
    """RawDataPreprocessorProjective2D::compute for one LaserMessage. Returns float32 [k, 4]."""
    r = np.ascontiguousarray(ranges, np.float32); assert len(r) == pp.n_beams
    out = np.empty((max(pp.n_beams, 1), 4), np.float32)
    k = lib().lsmo_preprocess_scan_f(C.byref(pp), r.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert k >= 0, k
    return out[:k].copy()
