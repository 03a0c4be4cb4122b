"""Host-side mirror of the reference's plugin surface for the scan-matching hot path, over the C ABI.

Class and method names follow the reference so that tests read like its own drivers:

* ``CorrespondenceFinderProjective2f`` / ``CorrespondenceFinderKDTree2D`` -- ``setFixed``, ``setMoving``,
  ``setLocalMapInSensor``, ``compute`` (apps/visual_test_correspondence_finder_projective_2d.cpp:74-79;
  parameters of registration/correspondence_finder_projective_2d.h:16-26, ..._kd_tree_2d.h:23-34).
* ``AlignerSliceProcessorLaser2D`` / ``...WithSensor`` (registration/aligner_slice_processor_laser_2d.h:7-42;
  config fields configurations/stage_segway_double_config_MULTI.json:160-188).
* ``MultiAligner2D`` -- ``param_slice_processors``, ``setFixed``, ``setMoving``, ``setMovingInFixed``, ``compute``,
  ``movingInFixed``, ``iterationStats`` (apps/visual_test_aligner_2d.cpp:123-156), plus ``compute_batch`` for the
  loop-closure / relocalisation sweeps the reference runs as a sequential loop (MULTI.json:964-986).

Errors: the reference throws ``std::runtime_error`` on missing inputs
(registration/correspondence_finder_projective_2d.cpp:21-31); here that is ``RuntimeError``; device/ABI failures raise
``Lsm2dError``.  All compute happens in the HIP library; nothing here falls back to a CPU path.
"""
from __future__ import annotations

import ctypes as C
import weakref
import dataclasses
import math
from typing import Optional, Sequence

import numpy as np

from . import _capi
from ._capi import (FINDER_NN, FINDER_PROJECTIVE, ROBUST_CAUCHY, ROBUST_NONE, AlignerParams, Batch, Correspondence,
                    IterationStats, Lsm2dError, Prior, Projector, SliceParams, check)

STATUS_NAMES = {0: "Success", 1: "NotEnoughCorrespondences", 2: "NotEnoughInliers", 3: "SingularH"}


class Context:
    """One device + one HIP stream (lsm2d_context)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, kernel_timing: bool = True):
        """kernel_timing: record HIP events around the hot-path launches so that ``last_kernel_ms()`` / ``BatchResult.kernel_ms``
        work (what the tests and bench.py want; the library's own default is off: the events cost a latency-critical caller
        ~20 % of a tracker step)."""
        self._lib = _capi.load()
        h = C.c_void_p()
        rc = self._lib.lsm2d_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise Lsm2dError(rc, "lsm2d_create", self._lib.lsm2d_last_error(None).decode())
        self._h = h
        self._sets = weakref.WeakSet()      # the cloud sets living on this context: close() destroys them first (a set outliving its context would touch freed memory)
        self.device = device
        self.kernel_timing = bool(kernel_timing)
        if kernel_timing:
            check(self._lib.lsm2d_set_option(self._h, b"kernel_timing", 1), "lsm2d_set_option", self._h)

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        check(self._lib.lsm2d_synchronize(self._h), "lsm2d_synchronize", self._h)

    def set_option(self, key: str, value: int):
        """e.g. ``set_option("align_path", 2)``: 0 automatic, 1 one workgroup per alignment, 2 split over many workgroups,
        3 two projective slices side by side in one workgroup."""
        if key == "kernel_timing":
            self.kernel_timing = bool(value)
        check(self._lib.lsm2d_set_option(self._h, key.encode(), int(value)), "lsm2d_set_option", self._h)

    def get_option(self, key: str) -> int:
        """Reads a knob back; ``"last_align_path"`` tells which kernels the latest ``align_batch`` ran (1, 2 or 3)."""
        v = C.c_int64()
        check(self._lib.lsm2d_get_option(self._h, key.encode(), C.byref(v)), "lsm2d_get_option", self._h)
        return int(v.value)

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        check(self._lib.lsm2d_last_kernel_ms(self._h, C.byref(ms)), "lsm2d_last_kernel_ms", self._h)
        return ms.value

    def close(self):
        if getattr(self, "_h", None):
            for cs in list(getattr(self, "_sets", ())):
                cs.close()
            self._lib.lsm2d_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CloudSet:
    """Device-resident ragged set of PointNormal2fVectorCloud (lsm2d_cloudset)."""

    def __init__(self, ctx: Context, points, offsets=None):
        self._ctx = ctx
        self._lib = ctx._lib
        h = C.c_void_p()
        if hasattr(points, "data_ptr"):          # a torch tensor already on ctx's device
            if not points.is_cuda or not points.is_contiguous() or points.dtype.itemsize != 4 or points.shape[-1] != 4:
                raise ValueError("device points must be a contiguous float32 [N, 4] tensor on the GPU")
            total = int(points.shape[0])
            offs = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
            n_clouds = 1 if offs is None else len(offs) - 1
            _producer_stream_wait(points)
            rc = self._lib.lsm2d_cloudset_create_from_device(
                ctx.handle, C.c_void_p(points.data_ptr()), None if offs is None else offs.ctypes.data_as(C.c_void_p),
                n_clouds, total, C.byref(h))
        else:
            pts = np.ascontiguousarray(points, np.float32)
            if pts.ndim != 2 or pts.shape[1] != 4:
                raise ValueError("points must be [N, 4] (x, y, nx, ny)")
            total = len(pts)
            offs = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
            n_clouds = 1 if offs is None else len(offs) - 1
            rc = self._lib.lsm2d_cloudset_create(ctx.handle, pts.ctypes.data_as(C.c_void_p),
                                                 None if offs is None else offs.ctypes.data_as(C.c_void_p),
                                                 n_clouds, total, C.byref(h))
        check(rc, "lsm2d_cloudset_create", ctx.handle)
        self._h = h
        ctx._sets.add(self)
        self.n_clouds = n_clouds
        self.n_points = total
        self.counts = np.array([total], np.int64) if offs is None else np.diff(offs.astype(np.int64))

    @property
    def handle(self):
        return self._h

    @classmethod
    def reserved(cls, ctx: Context, capacity: int) -> "CloudSet":
        """One growable device cloud (the local map a tracker keeps merging into / a clipped scene)."""
        self = cls.__new__(cls)
        self._ctx, self._lib = ctx, ctx._lib
        h = C.c_void_p()
        check(ctx._lib.lsm2d_cloudset_create_reserved(ctx.handle, int(capacity), C.byref(h)), "lsm2d_cloudset_create_reserved", ctx.handle)
        self._h, self.n_clouds, self.n_points, self.counts, self.capacity = h, 1, 0, np.zeros(1, np.int64), int(capacity)
        ctx._sets.add(self)
        return self

    # n_points / counts of a reserved set can be "known to the device only" after an asynchronous clip / merge
    # (SceneClipperProjective2D / MergerProjective2D with asynchronous=True): reading them then asks the library, which
    # synchronises once.
    @property
    def n_points(self) -> int:
        self._resolve()
        return self._n_points

    @n_points.setter
    def n_points(self, v):
        self._n_points = int(v)

    @property
    def counts(self) -> np.ndarray:
        self._resolve()
        return self._counts

    @counts.setter
    def counts(self, v):
        self._counts = v

    def _resolve(self):
        if getattr(self, "_pending", False):
            n = int(self._lib.lsm2d_cloudset_cloud_size(self._h, 0))
            self._pending = False
            self._n_points = n; self._counts = np.array([n], np.int64)

    def _set_count(self, n: int):
        self._pending = False
        self._n_points = int(n); self._counts = np.array([int(n)], np.int64)

    def _set_pending(self):
        self._pending = True

    def upload(self, points):
        """Refill this single-cloud set in place (no allocation, no wait for the stream: the points are copied into the
        set's pinned staging buffer before the call returns)."""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 4)
        check(self._lib.lsm2d_cloudset_upload(self._h, pts.ctypes.data_as(C.c_void_p), len(pts)), "lsm2d_cloudset_upload", self._ctx.handle)
        self._set_count(len(pts))

    def download(self, cloud_index: int = 0) -> np.ndarray:
        cap = int(self.counts[cloud_index])
        out = np.empty((max(cap, 1), 4), np.float32); n = C.c_int64(0)
        check(self._lib.lsm2d_cloudset_download(self._h, cloud_index, out.ctypes.data_as(C.c_void_p), cap, C.byref(n)),
              "lsm2d_cloudset_download", self._ctx.handle)
        return out[: n.value].copy()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lsm2d_cloudset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _producer_stream_wait(tensor):
    """A device tensor handed to the C ABI is read on the CONTEXT's stream, which knows nothing about the stream that produced the
    tensor (torch's current stream): wait for that stream here, so a tensor computed a moment ago is read complete (include/lsm2d.h,
    ORDERING note).  A context created on torch's own stream would not need this; the wait is cheap when the stream is idle."""
    if getattr(tensor, "is_cuda", False):
        import torch
        torch.cuda.current_stream(tensor.device).synchronize()


def _as_cloudset(ctx: Context, cloud) -> CloudSet:
    return cloud if isinstance(cloud, CloudSet) else CloudSet(ctx, cloud)


@dataclasses.dataclass
class PointNormal2fProjectorPolar:
    """Parameters of the polar projector (apps/synthetic_scene_generator.cpp:69-75; MULTI.json:71-97)."""
    param_canvas_cols: int = 721
    param_angle_col_min: float = -math.pi
    param_angle_col_max: float = math.pi
    param_range_min: float = 0.3
    param_range_max: float = 20.0
    col_offset: float = 0.0

    def struct(self) -> Projector:
        return Projector(self.param_canvas_cols, self.param_angle_col_min, self.param_angle_col_max,
                         self.param_range_min, self.param_range_max, self.col_offset)

    def compute(self, ctx: Context, cloud, pose=(0.0, 0.0, 0.0), cloud_index: int = 0):
        """One z-buffer pass; ``pose`` maps cloud points into the camera frame (= camera_pose^-1).
        Returns (source_idx int32 [cols], depth float32 [cols], transformed float32 [cols, 4])."""
        cs = _as_cloudset(ctx, cloud)
        cols = self.param_canvas_cols
        src = np.empty(cols, np.int32); depth = np.empty(cols, np.float32); xy = np.empty((cols, 4), np.float32)
        pose = np.ascontiguousarray(pose, np.float32)
        pr = self.struct()
        check(ctx._lib.lsm2d_project(ctx.handle, C.byref(pr), cs.handle, cloud_index, pose.ctypes.data_as(C.c_void_p),
                                     src.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p),
                                     xy.ctypes.data_as(C.c_void_p)), "lsm2d_project", ctx.handle)
        return src, depth, xy


class _FinderBase:
    finder_kind = FINDER_PROJECTIVE

    def __init__(self, ctx: Context):
        self._ctx = ctx
        self._fixed = None
        self._moving = None
        self._local_map_in_sensor = np.zeros(3, np.float32)
        self._correspondences = np.zeros((0, 2), np.int32)
        self._fixed_index = 0
        self._moving_index = 0

    # CorrespondenceFinder_ base-class surface (registration/correspondence_finder_normal_2f.h:9-13)
    def setFixed(self, fixed, index: int = 0):
        self._fixed = _as_cloudset(self._ctx, fixed); self._fixed_index = index

    def setMoving(self, moving, index: int = 0):
        self._moving = _as_cloudset(self._ctx, moving); self._moving_index = index

    def setLocalMapInSensor(self, pose):
        self._local_map_in_sensor = np.ascontiguousarray(pose, np.float32).reshape(3)

    def correspondences(self) -> np.ndarray:
        return self._correspondences

    def slice_params(self, **kw) -> SliceParams:
        raise NotImplementedError

    def _capacity(self) -> int:
        raise NotImplementedError

    def compute(self) -> np.ndarray:
        if self._fixed is None:
            raise RuntimeError(type(self).__name__ + "::compute| Missing fixed!")
        if self._moving is None:
            raise RuntimeError(type(self).__name__ + "::compute| Missing moving!")
        sp = self.slice_params()
        cap = max(self._capacity(), 1)
        out = np.empty((cap, 2), np.int32)
        n = C.c_int32(0)
        lib = self._ctx._lib
        check(lib.lsm2d_find_correspondences(self._ctx.handle, C.byref(sp), self._fixed.handle, self._fixed_index,
                                             self._moving.handle, self._moving_index,
                                             self._local_map_in_sensor.ctypes.data_as(C.c_void_p),
                                             out.ctypes.data_as(C.c_void_p), cap, C.byref(n)),
              "lsm2d_find_correspondences", self._ctx.handle)
        self._correspondences = out[:n.value].copy()
        return self._correspondences


class CorrespondenceFinderProjective2f(_FinderBase):
    """registration/correspondence_finder_projective_2d.{h,cpp}"""
    finder_kind = FINDER_PROJECTIVE

    def __init__(self, ctx: Context, projector: Optional[PointNormal2fProjectorPolar] = None,
                 point_distance: float = 0.5, normal_cos: float = 0.8):
        super().__init__(ctx)
        self.param_projector = projector
        self.param_point_distance = point_distance
        self.param_normal_cos = normal_cos

    def slice_params(self, **kw) -> SliceParams:
        if self.param_projector is None:
            raise RuntimeError("CorrespondenceFinderProjective2f::compute| Missing Projector")
        return make_slice_params(finder=FINDER_PROJECTIVE, projector=self.param_projector,
                                 point_distance=self.param_point_distance, normal_cos=self.param_normal_cos, **kw)

    def _capacity(self) -> int:
        return self.param_projector.param_canvas_cols


class CorrespondenceFinderKDTree2D(_FinderBase):
    """registration/correspondence_finder_kd_tree_2d.{h,cpp}.  ``search``: "exact" (LSM2D_FINDER_NN: exact nearest neighbour on a uniform
    grid; max_leaf_range / min_leaf_points have no meaning there) or "kdtree" (LSM2D_FINDER_KDTREE: the reference's own tree --
    KDTree2D(coordinates, max_leaf_range, min_leaf_points), .cpp:36-37 -- and its single-leaf descent, hence approximate).  The SRRG-side
    sibling (adapters/srrg) defaults to "kdtree"; this mirror keeps "exact" as its default so that existing drivers read unchanged."""

    def __init__(self, ctx: Context, max_distance_m: float = 1e-2, normal_cos: float = 0.8,
                 max_leaf_range: float = 1e-2, min_leaf_points: int = 20, search: str = "exact"):
        super().__init__(ctx)
        if search not in ("exact", "kdtree"):
            raise ValueError('search must be "exact" or "kdtree"')
        self.param_max_distance_m = max_distance_m
        self.param_normal_cos = normal_cos
        self.param_max_leaf_range = max_leaf_range      # honoured by search="kdtree"
        self.param_min_leaf_points = min_leaf_points
        self.search = search

    @property
    def finder_kind(self):
        return _capi.FINDER_KDTREE if self.search == "kdtree" else FINDER_NN

    def slice_params(self, **kw) -> SliceParams:
        return make_slice_params(finder=self.finder_kind, projector=PointNormal2fProjectorPolar(),
                                 max_distance=self.param_max_distance_m, normal_cos=self.param_normal_cos,
                                 kd_max_leaf_range=self.param_max_leaf_range, kd_min_leaf_points=self.param_min_leaf_points, **kw)

    def _capacity(self) -> int:
        return int(self._moving.counts[self._moving_index])


class CorrespondenceFinderNN2D(_FinderBase):
    """registration/correspondence_finder_nn_2d.{h,cpp}: distance-map finder (one grid lookup per query)."""
    finder_kind = _capi.FINDER_DISTMAP

    def __init__(self, ctx: Context, max_distance_m: float = 1.0, resolution: float = 5e-2, normal_cos: float = 0.8):
        super().__init__(ctx)
        self.param_max_distance_m = max_distance_m
        self.param_resolution = resolution
        self.param_normal_cos = normal_cos

    def slice_params(self, **kw) -> SliceParams:
        if self.param_resolution <= 0:
            raise RuntimeError("resolution must be > 0")               # correspondence_finder_nn_2d.cpp:11-14
        if self.param_max_distance_m < 0:
            raise RuntimeError("please set max_distance_m > 0")        # :15-18
        return make_slice_params(finder=_capi.FINDER_DISTMAP, projector=PointNormal2fProjectorPolar(),
                                 max_distance=self.param_max_distance_m, resolution=self.param_resolution,
                                 normal_cos=self.param_normal_cos, **kw)

    def _capacity(self) -> int:
        return int(self._moving.counts[self._moving_index])


def make_slice_params(finder=FINDER_PROJECTIVE, projector: Optional[PointNormal2fProjectorPolar] = None,
                      point_distance=0.5, normal_cos=0.8, max_distance=0.5, resolution=0.05,
                      robustifier=ROBUST_NONE, chi_threshold=0.05, min_num_correspondences=10,
                      sensor_in_robot=(0.0, 0.0, 0.0), kd_max_leaf_range=1e-2, kd_min_leaf_points=20) -> SliceParams:
    sp = SliceParams()
    sp.finder = finder
    sp.projector = (projector or PointNormal2fProjectorPolar()).struct()
    sp.point_distance, sp.normal_cos, sp.max_distance, sp.resolution = point_distance, normal_cos, max_distance, resolution
    sp.robustifier, sp.chi_threshold, sp.min_num_correspondences = robustifier, chi_threshold, min_num_correspondences
    sp.sensor_in_robot = (C.c_float * 3)(*[float(v) for v in sensor_in_robot])
    sp.kd_max_leaf_range, sp.kd_min_leaf_points = float(kd_max_leaf_range), int(kd_min_leaf_points)
    return sp


@dataclasses.dataclass
class RobustifierCauchy:
    """MULTI.json:153-158"""
    param_chi_threshold: float = 0.01


class AlignerSliceProcessorLaser2D:
    """registration/aligner_slice_processor_laser_2d.h:7-17 -- finder + SE2Plane2PlaneErrorFactor (+ robustifier)."""

    def __init__(self, finder: _FinderBase, robustifier: Optional[RobustifierCauchy] = None,
                 min_num_correspondences: int = 0, fixed_slice_name: str = "points", moving_slice_name: str = "points"):
        self.param_finder = finder
        self.param_robustifier = robustifier
        self.param_min_num_correspondences = min_num_correspondences
        self.param_fixed_slice_name = fixed_slice_name
        self.param_moving_slice_name = moving_slice_name
        self.sensor_in_robot = (0.0, 0.0, 0.0)

    def slice_params(self) -> SliceParams:
        rb = self.param_robustifier
        return self.param_finder.slice_params(
            robustifier=ROBUST_CAUCHY if rb else ROBUST_NONE,
            chi_threshold=rb.param_chi_threshold if rb else 0.0,
            min_num_correspondences=self.param_min_num_correspondences, sensor_in_robot=self.sensor_in_robot)


class AlignerSliceProcessorLaser2DWithSensor(AlignerSliceProcessorLaser2D):
    """registration/aligner_slice_processor_laser_2d.h:21-42: the estimate lives in the robot frame, the
    fixed scan in the sensor frame; ``sensor_in_robot`` comes from the tf Platform in the reference
    (apps/visual_test_aligner_2d.cpp:96-107)."""

    def __init__(self, finder, sensor_in_robot=(0.0, 0.0, 0.0), **kw):
        super().__init__(finder, **kw)
        self.sensor_in_robot = tuple(float(v) for v in sensor_in_robot)


@dataclasses.dataclass
class BatchResult:
    pose: np.ndarray          # [n, 3]  movingInFixed
    information: np.ndarray   # [n, 3, 3]
    status: np.ndarray        # [n]
    iterations: np.ndarray    # [n]
    stats: Optional[np.ndarray]  # structured [n, lsm2d_stats_capacity] or None
    kernel_ms: float
    kernel_clock_mhz: float = 0.0      # clock the chip held inside the k_align launch (in-kernel stamps; 0 when not timed / another kernel ran)
    workgroup_lifetime_ms: float = 0.0  # median lifetime of the stamped workgroups
    pairs: Optional[list] = None        # want_pairs: pairs[i][s] = int32 [k, 2] (fixed_idx, moving_idx) the aligner leaves in slice s of alignment i

    def status_names(self):
        return [STATUS_NAMES.get(int(s), str(int(s))) for s in self.status]

    def last_stats(self) -> np.ndarray:
        """Statistics of the last iteration each alignment started (structured [n])."""
        if self.stats is None:
            raise RuntimeError("compute_batch(..., want_stats=True) is needed for per-alignment statistics")
        idx = np.clip(self.iterations - 1, 0, self.stats.shape[1] - 1)
        return self.stats[np.arange(len(idx)), idx]

    def loop_closure_accept(self, relocalize_min_inliers: int = 500, relocalize_max_chi_inliers: float = 0.1,
                            relocalize_min_inliers_ratio: float = 0.8) -> np.ndarray:
        """Acceptance test MultiLoopDetectorBruteForce2D applies to every candidate after relocalize_aligner
        (configurations/stage_segway_double_config_MULTI.json:964-986): aligner succeeded, inliers >= min,
        chi_inliers / inliers <= max, inliers / correspondences >= ratio.  The relocaliser uses the same test with
        700 / 0.01 / 0.75 (MULTI.json:749-769).  Returns bool [n]."""
        st = self.last_stats()
        n_in = st["n_inliers"].astype(np.float64); n_c = np.maximum(st["n_correspondences"], 1).astype(np.float64)
        ok = (self.status == 0) & (st["n_inliers"] >= relocalize_min_inliers)
        ok &= st["chi_inliers"] / np.maximum(n_in, 1.0) <= relocalize_max_chi_inliers
        ok &= n_in / n_c >= relocalize_min_inliers_ratio
        return ok


STATS_DTYPE = np.dtype([("n_correspondences", np.int32), ("n_inliers", np.int32), ("n_outliers", np.int32),
                        ("chi_inliers", np.float32), ("chi_outliers", np.float32),
                        ("pair_digest_lo", np.uint32), ("pair_digest_hi", np.uint32)])


def pair_digests(stats: np.ndarray) -> np.ndarray:
    """uint64 digest of every iteration's correspondence set (lsm2d_iteration_stats.pair_digest_lo / _hi)."""
    return (stats["pair_digest_hi"].astype(np.uint64) << np.uint64(32)) | stats["pair_digest_lo"].astype(np.uint64)


class MultiAligner2D:
    """The upstream aligner as the reference drives it (apps/visual_test_aligner_2d.cpp:123-156), with the
    whole iteration loop running on the device."""

    def __init__(self, ctx: Context, max_iterations: int = 10, min_num_inliers: int = 10, damping: float = 0.0,
                 termination_chi_epsilon: float = 0.0):
        self._ctx = ctx
        self.param_max_iterations = max_iterations
        self.param_min_num_inliers = min_num_inliers
        self.param_damping = damping
        # the options both shipped aligners carry at their defaults (MULTI.json:606-610,627-630,704-708,729-731); semantics in include/lsm2d.h
        # (restated from the parameters' doc strings: the upstream class is not in the reference tree).  The termination criterion exists as
        # an epsilon on the relative decay of the total chi^2 (lsm2d.h); 0 = not set = max_iterations.
        self.param_enable_inlier_only_runs = False
        self.param_keep_only_inlier_correspondences = False
        self.param_termination_chi_epsilon = termination_chi_epsilon
        self.store_correspondences = False      # compute() also fetches what the reference leaves in slice->correspondences()
        self.param_slice_processors: list[AlignerSliceProcessorLaser2D] = []
        self._fixed = {}
        self._moving = {}
        self._moving_in_fixed = np.zeros(3, np.float32)
        self._prior = None
        self._result: Optional[BatchResult] = None

    # --- single-alignment surface ---------------------------------------------------------------
    def setFixed(self, container: dict):
        """``container`` maps slice names to clouds (the PropertyContainer of the reference)."""
        self._fixed = {k: _as_cloudset(self._ctx, v) for k, v in container.items()}

    def setMoving(self, container: dict):
        self._moving = {k: _as_cloudset(self._ctx, v) for k, v in container.items()}

    def setMovingInFixed(self, pose):
        self._moving_in_fixed = np.ascontiguousarray(pose, np.float32).reshape(3)

    def setPrior(self, z, omega):
        """Odometry-prior cue (AlignerSliceOdom2DPrior, MULTI.json:402-422)."""
        self._prior = (np.asarray(z, np.float32).reshape(3), np.asarray(omega, np.float32).reshape(3, 3))

    def compute(self):
        if not self.param_slice_processors:
            raise RuntimeError("MultiAligner2D::compute| no slice processors")
        fixed = [self._fixed[s.param_fixed_slice_name] for s in self.param_slice_processors]
        moving = [self._moving[s.param_moving_slice_name] for s in self.param_slice_processors]
        prior = None if self._prior is None else [self._prior]
        self._result = self.compute_batch(fixed, moving, self._moving_in_fixed[None, :], priors=prior, want_stats=True,
                                          want_pairs=self.store_correspondences)
        return self.status()

    def correspondences(self, slice_index: int = 0) -> np.ndarray:
        """slice->correspondences() after compute() (apps/visual_test_aligner_2d.cpp:129-143): int32 [k, 2] (fixed_idx, moving_idx).
        Needs ``store_correspondences = True`` before compute(): the device loop keeps no pair lists, they cost a finder pass per slice."""
        if self._result is None or self._result.pairs is None:
            raise RuntimeError("MultiAligner2D::correspondences| set store_correspondences = True before compute()")
        return self._result.pairs[0][slice_index]

    def movingInFixed(self) -> np.ndarray:
        return self._result.pose[0]

    def informationMatrix(self) -> np.ndarray:
        return self._result.information[0]

    def status(self) -> int:
        return int(self._result.status[0])

    def iterationStats(self):
        r = self._result
        return r.stats[0][: int(r.iterations[0])]

    # --- batched surface ---------------------------------------------------------------------------
    def compute_batch(self, fixed: Sequence[CloudSet], moving: Sequence[CloudSet], init_poses, priors=None,
                      fixed_index=None, moving_index=None, want_stats: bool = False, want_pairs: bool = False) -> BatchResult:
        """``fixed[s]`` / ``moving[s]``: cloud set of slice ``s`` (one cloud = shared by the batch, else one per
        alignment or chosen through ``*_index[s][i]``).  ``init_poses``: [n, 3].  want_pairs: also the correspondences the aligner
        leaves in its slices (lsm2d_align_batch_pairs: one finder pass per alignment and slice after the aligner kernel)."""
        ctx, lib = self._ctx, self._ctx._lib
        slices = self.param_slice_processors
        ns = len(slices)
        b, keep = self._batch(fixed, moving, init_poses, priors, fixed_index, moving_index)
        n = b.n_alignments; moving_sets = keep[1]
        ap = AlignerParams(self.param_max_iterations, self.param_min_num_inliers, self.param_damping, self.param_termination_chi_epsilon,
                           1 if self.param_enable_inlier_only_runs else 0, 1 if self.param_keep_only_inlier_correspondences else 0)
        pose = np.empty((n, 3), np.float32); H = np.empty((n, 9), np.float32)
        status = np.empty(n, np.int32); its = np.empty(n, np.int32)
        stats = np.zeros((n, int(lib.lsm2d_stats_capacity(C.byref(ap)))), STATS_DTYPE) if want_stats else None
        pairs = None
        if want_pairs and n:
            cap = 1
            for s_, m_ in zip(slices, moving_sets):
                sp_ = s_.slice_params()
                cap = max(cap, sp_.projector.canvas_cols if sp_.finder == FINDER_PROJECTIVE else int(max(m_.counts)) if len(m_.counts) else 1)
            pbuf = np.empty((n, ns, cap, 2), np.int32); pcnt = np.zeros((n, ns), np.int32)
            check(lib.lsm2d_align_batch_pairs(ctx.handle, C.byref(ap), C.byref(b), pose.ctypes.data_as(C.c_void_p),
                                              H.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p), its.ctypes.data_as(C.c_void_p),
                                              stats.ctypes.data_as(C.c_void_p) if want_stats else None,
                                              pbuf.ctypes.data_as(C.c_void_p), cap, pcnt.ctypes.data_as(C.c_void_p)),
                  "lsm2d_align_batch_pairs", ctx.handle)
            pairs = [[pbuf[i, s_, : pcnt[i, s_]].copy() for s_ in range(ns)] for i in range(n)]
        else:
            check(lib.lsm2d_align_batch(ctx.handle, C.byref(ap), C.byref(b), pose.ctypes.data_as(C.c_void_p),
                                        H.ctypes.data_as(C.c_void_p), status.ctypes.data_as(C.c_void_p),
                                        its.ctypes.data_as(C.c_void_p),
                                        stats.ctypes.data_as(C.c_void_p) if want_stats else None),
                  "lsm2d_align_batch", ctx.handle)
        timed = bool(n and ctx.kernel_timing) and not pairs      # (the finder passes behind a pairs call overwrite the timed events)
        return BatchResult(pose, H.reshape(n, 3, 3), status, its, stats, ctx.last_kernel_ms() if timed else 0.0,
                           ctx.get_option("last_kernel_clock_khz") * 1e-3 if timed else 0.0,
                           ctx.get_option("last_workgroup_lifetime_ns") * 1e-6 if timed else 0.0, pairs)


def _batch_method(self, fixed, moving, init_poses, priors=None, fixed_index=None, moving_index=None):
    """lsm2d_batch descriptor of a call (and everything it points to, to be kept alive by the caller)"""
    ctx = self._ctx
    slices = self.param_slice_processors
    ns = len(slices)
    x0 = np.array(init_poses, dtype=np.float32, order="C", copy=True).reshape(-1, 3)      # the descriptor's OWN copy (round-4 advisor: PreparedBatch.set_init_poses wrote into the caller's array)
    n = len(x0)
    sp = (SliceParams * ns)(*[s.slice_params() for s in slices])
    fixed_sets = [_as_cloudset(ctx, f) for f in fixed]          # keep host-array uploads alive for the duration of the call
    moving_sets = [_as_cloudset(ctx, m) for m in moving]
    fx = (C.c_void_p * ns)(*[f.handle.value for f in fixed_sets])
    mv = (C.c_void_p * ns)(*[m.handle.value for m in moving_sets])
    b = Batch()
    b.n_alignments, b.n_slices = n, ns
    b.slices = sp
    b.fixed = C.cast(fx, C.POINTER(C.c_void_p)); b.moving = C.cast(mv, C.POINTER(C.c_void_p))
    keep = [fixed_sets, moving_sets, sp, fx, mv, x0]
    if fixed_index is not None:
        fi = np.ascontiguousarray(fixed_index, np.int32).reshape(ns, n); keep.append(fi)
        b.fixed_index = fi.ctypes.data_as(C.POINTER(C.c_int32))
    if moving_index is not None:
        mi = np.ascontiguousarray(moving_index, np.int32).reshape(ns, n); keep.append(mi)
        b.moving_index = mi.ctypes.data_as(C.POINTER(C.c_int32))
    b.init_pose = x0.ctypes.data_as(C.POINTER(C.c_float))
    if priors is not None:
        pr = (Prior * n)()
        for i, (z, om) in enumerate(priors):
            pr[i].z = (C.c_float * 3)(*np.asarray(z, np.float32).ravel())
            pr[i].omega = (C.c_float * 9)(*np.asarray(om, np.float32).ravel())
        b.prior = pr; keep.append(pr)
    return b, keep


def _estimate_work_method(self, fixed, moving, init_poses, fixed_index=None, moving_index=None) -> np.ndarray:
    """lsm2d_estimate_work: per alignment, what it will cost relative to the others (chunks of the moving cloud its first iteration streams),
    without running it -- what a sweep sharded over devices or ranks balances its shards by (distributed.shard_by_work).  int32 [n]."""
    b, keep = self._batch(fixed, moving, init_poses, None, fixed_index, moving_index)
    work = np.empty(b.n_alignments, np.int32)
    check(self._ctx._lib.lsm2d_estimate_work(self._ctx.handle, C.byref(b), work.ctypes.data_as(C.c_void_p)), "lsm2d_estimate_work", self._ctx.handle)
    return work


class PreparedBatch:
    """A batch whose descriptor, parameters and result arrays are built ONCE and handed to lsm2d_align_batch again and again -- what a host loop in the
    reference's own language does with its vectors (a candidate sweep re-aligns the same sets from new poses; the bench times the call, not the
    interpreter's marshalling of it: ~25 us of ctypes / numpy per call otherwise).  ``set_init_poses`` overwrites the start poses in place; every
    ``run()`` overwrites the result arrays of the BatchResult it returns: the results of successive runs SHARE their arrays (``run(copy=True)`` hands back
    arrays of their own).  The library keeps the placement of a batch it has seen before -- a run with unchanged sets and start poses launches no estimate
    (``Context.get_option("last_cull_estimate")`` tells)."""

    def __init__(self, aligner, fixed, moving, init_poses, priors=None, fixed_index=None, moving_index=None, want_stats: bool = False):
        self._aligner = aligner
        self._ctx = aligner._ctx
        self._b, self._keep = aligner._batch(fixed, moving, init_poses, priors, fixed_index, moving_index)
        self._x0 = self._keep[5]
        n = self._b.n_alignments
        self._ap = AlignerParams(aligner.param_max_iterations, aligner.param_min_num_inliers, aligner.param_damping, aligner.param_termination_chi_epsilon,
                                 1 if aligner.param_enable_inlier_only_runs else 0, 1 if aligner.param_keep_only_inlier_correspondences else 0)
        lib = self._ctx._lib
        self.pose = np.empty((n, 3), np.float32); self._H = np.empty((n, 9), np.float32)
        self.status = np.empty(n, np.int32); self.iterations = np.empty(n, np.int32)
        self.stats = np.zeros((n, int(lib.lsm2d_stats_capacity(C.byref(self._ap)))), STATS_DTYPE) if want_stats else None
        self._args = (self._ctx.handle, C.byref(self._ap), C.byref(self._b), self.pose.ctypes.data_as(C.c_void_p), self._H.ctypes.data_as(C.c_void_p),
                      self.status.ctypes.data_as(C.c_void_p), self.iterations.ctypes.data_as(C.c_void_p),
                      self.stats.ctypes.data_as(C.c_void_p) if want_stats else None)
        self._fn = lib.lsm2d_align_batch

    def set_init_poses(self, init_poses) -> None:
        self._x0[...] = np.asarray(init_poses, np.float32).reshape(self._x0.shape)

    def begin(self) -> None:
        """lsm2d_align_batch_begin: queue the batch and return; ``wait()`` hands its results over.  At most two batches of a context may be in flight, waited for in
        the order they were begun -- two PreparedBatch objects alternating (each keeps its own result arrays)."""
        if getattr(self, "_pending", None) is not None:
            raise RuntimeError("PreparedBatch.begin: this batch is in flight already")
        h = C.c_void_p()
        check(self._ctx._lib.lsm2d_align_batch_begin(self._ctx.handle, C.byref(self._ap), C.byref(self._b), 1 if self.stats is not None else 0, C.byref(h)),
              "lsm2d_align_batch_begin", self._ctx.handle)
        self._pending = h

    def wait(self, copy: bool = False) -> BatchResult:
        if getattr(self, "_pending", None) is None:
            raise RuntimeError("PreparedBatch.wait: nothing in flight")
        h, self._pending = self._pending, None
        ctx = self._ctx
        check(ctx._lib.lsm2d_align_batch_wait(h, self._args[3], self._args[4], self._args[5], self._args[6], self._args[7]), "lsm2d_align_batch_wait", ctx.handle)
        n = self._b.n_alignments
        timed = bool(n and ctx.kernel_timing)
        c = (lambda a: None if a is None else a.copy()) if copy else (lambda a: a)
        return BatchResult(c(self.pose), c(self._H).reshape(n, 3, 3), c(self.status), c(self.iterations), c(self.stats), ctx.last_kernel_ms() if timed else 0.0,
                           ctx.get_option("last_kernel_clock_khz") * 1e-3 if timed else 0.0,
                           ctx.get_option("last_workgroup_lifetime_ns") * 1e-6 if timed else 0.0, None)

    def run(self, copy: bool = False) -> BatchResult:
        ctx = self._ctx
        check(self._fn(*self._args), "lsm2d_align_batch", ctx.handle)
        n = self._b.n_alignments
        timed = bool(n and ctx.kernel_timing)
        c = (lambda a: None if a is None else a.copy()) if copy else (lambda a: a)
        return BatchResult(c(self.pose), c(self._H).reshape(n, 3, 3), c(self.status), c(self.iterations), c(self.stats), ctx.last_kernel_ms() if timed else 0.0,
                           ctx.get_option("last_kernel_clock_khz") * 1e-3 if timed else 0.0,
                           ctx.get_option("last_workgroup_lifetime_ns") * 1e-6 if timed else 0.0, None)


def run_pipelined(batches, copy: bool = True):
    """A queue of prepared batches with two in flight: ``begin(k)`` ; ``wait(k - 1)`` -- each asynchronously begun batch launches on its lane's own stream, so the
    younger launch's workgroups fill the slots the older one's tail leaves free (configs[1]: 0.70 ms per batch against 0.77 one at a time).  Yields the BatchResults in
    the order of the batches; every result is what ``run()`` of that batch returns, bit for bit.  Consecutive entries must be DIFFERENT PreparedBatch objects (each
    keeps its own result arrays; the same object may come again once its previous run has been yielded, i.e. two entries later)."""
    prev = None
    for b in batches:
        b.begin()
        if prev is not None:
            yield prev.wait(copy=copy)
        prev = b
    if prev is not None:
        yield prev.wait(copy=copy)


def _prepare_batch_method(self, fixed, moving, init_poses, priors=None, fixed_index=None, moving_index=None, want_stats: bool = False) -> PreparedBatch:
    """compute_batch's arguments, marshalled once: ``prepare_batch(...).run()`` == ``compute_batch(...)`` (tests), call after call."""
    return PreparedBatch(self, fixed, moving, init_poses, priors, fixed_index, moving_index, want_stats)


MultiAligner2D._batch = _batch_method
MultiAligner2D.estimate_work = _estimate_work_method
MultiAligner2D.prepare_batch = _prepare_batch_method


def linearize(ctx: Context, slice_params: SliceParams, fixed, moving, correspondences, pose,
              fixed_index: int = 0, moving_index: int = 0):
    """SE2Plane2PlaneErrorFactor over a correspondence vector: returns (H [3,3], b [3], IterationStats)."""
    fx, mv = _as_cloudset(ctx, fixed), _as_cloudset(ctx, moving)
    corr = np.ascontiguousarray(correspondences, np.int32).reshape(-1, 2)
    pose = np.ascontiguousarray(pose, np.float32)
    H = np.empty(9, np.float32); b = np.empty(3, np.float32); st = IterationStats()
    check(ctx._lib.lsm2d_linearize(ctx.handle, C.byref(slice_params), fx.handle, fixed_index, mv.handle, moving_index,
                                   corr.ctypes.data_as(C.c_void_p), len(corr), pose.ctypes.data_as(C.c_void_p),
                                   H.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), C.byref(st)),
          "lsm2d_linearize", ctx.handle)
    return H.reshape(3, 3), b, st


class SceneClipperProjective2D:
    """mapping/scene_clipper_projective_2d.{h,cpp}: keeps what the sensor sees of the local map, at most one point per projector
    column, in the robot frame; ``voxelize_resolution`` > 0 additionally voxelises the clipped cloud (.cpp:36-48; both shipped configs
    use 0, MULTI.json:673-683).  The clipped scene stays on the device (a reserved CloudSet) and is what the tracker hands to the
    aligner as ``moving``."""

    def __init__(self, ctx: Context, projector: Optional[PointNormal2fProjectorPolar] = None, voxelize_resolution: float = 0.1,
                 asynchronous: bool = False):
        self._ctx = ctx
        self.param_projector = projector
        self.param_voxelize_resolution = voxelize_resolution
        # asynchronous: compute() only queues the work -- nothing is copied back (source_indices stays empty) and the clipped
        # set's size is fetched when somebody reads it; the aligner and the merger take such a set as it is
        self.asynchronous = asynchronous
        self._full_scene = None
        self._clipped = None
        self._robot_in_local_map = np.zeros(3, np.float32)
        self._sensor_in_robot = np.zeros(3, np.float32)
        self.source_indices = np.zeros(0, np.int32)

    def setFullScene(self, scene):
        self._full_scene = _as_cloudset(self._ctx, scene)

    def setClippedSceneInRobot(self, clipped: CloudSet):
        self._clipped = clipped

    def setRobotInLocalMap(self, pose):
        self._robot_in_local_map = np.ascontiguousarray(pose, np.float32).reshape(3)

    def setSensorInRobot(self, pose):
        self._sensor_in_robot = np.ascontiguousarray(pose, np.float32).reshape(3)

    def compute(self) -> CloudSet:
        if self._full_scene is None:
            raise RuntimeError("SceneClipperProjective2D::compute| missing local OR global scene")
        if self.param_projector is None:
            raise RuntimeError("SceneClipperProjective2D::compute| Missing Projector")
        cols = self.param_projector.param_canvas_cols
        if self._clipped is None:
            self._clipped = CloudSet.reserved(self._ctx, cols)
        pr = self.param_projector.struct()
        vox = float(self.param_voxelize_resolution)
        lib = self._ctx._lib
        args = (self._ctx.handle, C.byref(pr), self._full_scene.handle, 0, self._robot_in_local_map.ctypes.data_as(C.c_void_p),
                self._sensor_in_robot.ctypes.data_as(C.c_void_p))

        def call(out_n, out_src):      # voxelize_resolution > 0: the branch of scene_clipper_projective_2d.cpp:36-48 (no source indices)
            if vox > 0:
                return check(lib.lsm2d_clip_scene_voxelized(*args, vox, self._clipped.handle, out_n, None), "lsm2d_clip_scene_voxelized", self._ctx.handle)
            return check(lib.lsm2d_clip_scene(*args, self._clipped.handle, out_n, out_src), "lsm2d_clip_scene", self._ctx.handle)

        if self.asynchronous:
            call(None, None)
            self._clipped._set_pending()
            self.source_indices = np.zeros(0, np.int32)
            return self._clipped
        n = C.c_int32(0); src = np.empty(cols, np.int32)
        call(C.byref(n), src.ctypes.data_as(C.c_void_p))
        self._clipped._set_count(n.value)
        self.source_indices = src[: n.value].copy() if not vox > 0 else np.zeros(0, np.int32)
        return self._clipped


class MergerProjective2D:
    """mapping/merger_projective_2d.{h,cpp}: folds a measurement into the device-resident scene, in place."""

    def __init__(self, ctx: Context, projector: Optional[PointNormal2fProjectorPolar] = None, merge_threshold: float = 0.2,
                 asynchronous: bool = False):
        self._ctx = ctx
        self.param_projector = projector
        self.param_merge_threshold = merge_threshold
        self.asynchronous = asynchronous      # compute() only queues the merge; the scene's new size is fetched when it is read
        self._scene = None
        self._measurement = None
        self._measurement_in_scene = np.zeros(3, np.float32)
        self.counts = (0, 0, 0)      # new, merged, replaced

    def setScene(self, scene: CloudSet):
        self._scene = scene

    def setMeasurement(self, measurement, index: int = 0):
        self._measurement = _as_cloudset(self._ctx, measurement); self._measurement_index = index

    def setMeasurementInScene(self, pose):
        self._measurement_in_scene = np.ascontiguousarray(pose, np.float32).reshape(3)

    def compute(self) -> int:
        if self.param_projector is None:
            raise RuntimeError("MergerProjective2D::compute| Missing Projector")
        if self._scene is None or self._measurement is None:
            raise RuntimeError("MergerProjective2D::compute| missing scene or measurement")
        pr = self.param_projector.struct()
        if self.asynchronous:
            check(self._ctx._lib.lsm2d_merge_scene(self._ctx.handle, C.byref(pr), self._scene.handle, self._measurement.handle,
                                                   self._measurement_index, self._measurement_in_scene.ctypes.data_as(C.c_void_p),
                                                   float(self.param_merge_threshold), None, None),
                  "lsm2d_merge_scene", self._ctx.handle)
            self._scene._set_pending()
            self.counts = (0, 0, 0)
            return -1
        size = C.c_int32(0); counts = (C.c_int32 * 3)()
        check(self._ctx._lib.lsm2d_merge_scene(self._ctx.handle, C.byref(pr), self._scene.handle, self._measurement.handle,
                                               self._measurement_index, self._measurement_in_scene.ctypes.data_as(C.c_void_p),
                                               float(self.param_merge_threshold), C.byref(size), counts),
              "lsm2d_merge_scene", self._ctx.handle)
        self._scene._set_count(size.value)
        self.counts = tuple(counts)
        return size.value


    def compute_all(self, measurements: Sequence, poses, indices: Optional[Sequence[int]] = None) -> int:
        """Merges several measurements, each at its own pose in the scene, in the given order -- what one setMeasurement /
        setMeasurementInScene / compute() round per measurement does, as ONE call (lsm2d_merge_scenes: one launch when the scene and
        the measurements are small).  Returns the scene's new size (-1 when asynchronous); ``counts`` then holds one triple per measurement."""
        if self.param_projector is None:
            raise RuntimeError("MergerProjective2D::compute| Missing Projector")
        if self._scene is None or not len(measurements):
            raise RuntimeError("MergerProjective2D::compute| missing scene or measurement")
        sets = [_as_cloudset(self._ctx, m) for m in measurements]
        n = len(sets)
        handles = (C.c_void_p * n)(*[s_.handle for s_ in sets])
        p = np.ascontiguousarray(poses, np.float32).reshape(n, 3)
        idx = None if indices is None else (C.c_int32 * n)(*[int(i) for i in indices])      # which cloud of each (multi-cloud) set
        pr = self.param_projector.struct()
        if self.asynchronous:
            check(self._ctx._lib.lsm2d_merge_scenes(self._ctx.handle, C.byref(pr), self._scene.handle, n, handles, idx, p.ctypes.data_as(C.c_void_p),
                                                    float(self.param_merge_threshold), None, None), "lsm2d_merge_scenes", self._ctx.handle)
            self._scene._set_pending()
            self.counts = (0, 0, 0)
            return -1
        size = C.c_int32(0); counts = (C.c_int32 * (3 * n))()
        check(self._ctx._lib.lsm2d_merge_scenes(self._ctx.handle, C.byref(pr), self._scene.handle, n, handles, idx, p.ctypes.data_as(C.c_void_p),
                                                float(self.param_merge_threshold), C.byref(size), counts), "lsm2d_merge_scenes", self._ctx.handle)
        self._scene._set_count(size.value)
        self.counts = [tuple(counts[3 * k:3 * k + 3]) for k in range(n)]
        return size.value


def _data_pointer(a):
    return C.c_void_p(a.data_ptr()) if hasattr(a, "data_ptr") else a.ctypes.data_as(C.c_void_p)


class RawDataPreprocessorProjective2D:
    """sensor_processing/raw_data_preprocessor_projective_2d.{h,cpp}: LaserMessage ranges -> PointNormal2fVectorCloud
    (polar unprojection, sliding-window normals, voxelisation), batched over scans; the clouds stay on the device."""

    def __init__(self, ctx: Context, range_min: float = 0.0, range_max: float = 1000.0, voxelize_resolution: float = 0.02,
                 normal_point_distance: float = 0.3, normal_min_points: int = 5):
        self._ctx = ctx
        self.param_range_min = range_min                    # .h:39
        self.param_range_max = range_max                    # .h:40
        self.param_voxelize_resolution = voxelize_resolution  # .h:41-45
        self.param_normal_point_distance = normal_point_distance   # NormalComputator1DSlidingWindow (MULTI.json:845-853)
        self.param_normal_min_points = normal_min_points
        self._msg = None
        self._meas: Optional[CloudSet] = None

    def setRawData(self, ranges, angle_min: float, angle_max: float, range_min: float = 0.0, range_max: float = float("inf")) -> bool:
        """One LaserMessage (ranges [n_beams]) or a batch of them ([n_scans, n_beams]) with the message's own limits.  A torch tensor is
        taken as it is: on the context's GPU the ranges are read in place, in pinned host memory they are copied from directly."""
        if hasattr(ranges, "data_ptr"):
            r = ranges if ranges.dim() == 2 else ranges[None, :]
            if r.dtype.itemsize != 4 or not r.dtype.is_floating_point or not r.is_contiguous():
                raise ValueError("tensor ranges must be contiguous float32")
        else:
            r = np.ascontiguousarray(ranges, np.float32)
            if r.ndim == 1:
                r = r[None, :]
        if r.ndim != 2 or r.shape[1] < 1:
            raise RuntimeError("RawDataPreprocessorProjective2D::setMeasurement|measurement is not set")     # .cpp:54-57
        self._msg = (r, float(angle_min), float(angle_max), float(range_min), float(range_max))
        return True

    def compute(self) -> CloudSet:
        if self._msg is None:
            raise RuntimeError("RawDataPreprocessorProjective2D::compute| no raw data")
        r, a0, a1, m_rmin, m_rmax = self._msg
        pp = _capi.Preprocessor(r.shape[1], a0, a1, max(m_rmin, self.param_range_min), min(m_rmax, self.param_range_max),   # .cpp:83-84
                                self.param_normal_point_distance, self.param_normal_min_points, self.param_voxelize_resolution)
        h = C.c_void_p()
        _producer_stream_wait(r)
        check(self._ctx._lib.lsm2d_preprocess_scans(self._ctx.handle, C.byref(pp), _data_pointer(r), r.shape[0], C.byref(h)),
              "lsm2d_preprocess_scans", self._ctx.handle)
        cs = CloudSet.__new__(CloudSet)
        cs._ctx, cs._lib, cs._h = self._ctx, self._ctx._lib, h
        self._ctx._sets.add(cs)
        cs.n_clouds = r.shape[0]
        cs.n_points = int(self._ctx._lib.lsm2d_cloudset_num_points(h))
        sizes = np.empty(r.shape[0], np.int32)
        check(min(self._ctx._lib.lsm2d_cloudset_cloud_sizes(h, sizes.ctypes.data_as(C.c_void_p), len(sizes)), 0), "lsm2d_cloudset_cloud_sizes", self._ctx.handle)
        cs.counts = sizes.astype(np.int64)
        self._meas = cs
        return cs

    def refill(self, out: CloudSet) -> CloudSet:
        """The streaming form: the batch of messages set with setRawData goes into ``out`` -- a set an earlier ``compute()`` made for the same number of scans
        and beams -- with no allocation and no wait (lsm2d_preprocess_scans_refill): the ranges are fetched by an asynchronous copy (keep them untouched until
        the batch that reads ``out`` has been waited for), the clouds' sizes stay on the device.  Same kernel, same bits as ``compute()``."""
        if self._msg is None:
            raise RuntimeError("RawDataPreprocessorProjective2D::compute| no raw data")
        r, a0, a1, m_rmin, m_rmax = self._msg
        pp = _capi.Preprocessor(r.shape[1], a0, a1, max(m_rmin, self.param_range_min), min(m_rmax, self.param_range_max),
                                self.param_normal_point_distance, self.param_normal_min_points, self.param_voxelize_resolution)
        _producer_stream_wait(r)
        check(self._ctx._lib.lsm2d_preprocess_scans_refill(self._ctx.handle, C.byref(pp), _data_pointer(r), r.shape[0], out.handle),
              "lsm2d_preprocess_scans_refill", self._ctx.handle)
        self._meas = out
        return out

    def compute_into(self, out: CloudSet) -> CloudSet:
        """The live tracker's form: the ONE scan set with setRawData goes into the reserved set ``out`` -- no allocation,
        nothing waits; the cloud's size stays on the device until it is read (lsm2d_preprocess_scan_into)."""
        if self._msg is None:
            raise RuntimeError("RawDataPreprocessorProjective2D::compute| no raw data")
        r, a0, a1, m_rmin, m_rmax = self._msg
        if r.shape[0] != 1:
            raise ValueError("compute_into takes one scan")
        pp = _capi.Preprocessor(r.shape[1], a0, a1, max(m_rmin, self.param_range_min), min(m_rmax, self.param_range_max),
                                self.param_normal_point_distance, self.param_normal_min_points, self.param_voxelize_resolution)
        if hasattr(r, "data_ptr") and r.is_cuda:
            raise ValueError("compute_into stages the scan from host memory")
        check(self._ctx._lib.lsm2d_preprocess_scan_into(self._ctx.handle, C.byref(pp), _data_pointer(r), out.handle),
              "lsm2d_preprocess_scan_into", self._ctx.handle)
        out._set_pending()
        self._meas = out
        return out
