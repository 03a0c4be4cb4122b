"""ctypes binding of the C ABI in include/lsm2d.h (liblsm2d_hip.so).

There is no CPU fallback: importing works everywhere (so the symbol table can be checked without a
GPU), but creating a context raises unless a HIP device is present, and a missing library raises
at load time.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os

from . import build as _build

SUCCESS, NOT_ENOUGH_CORRESPONDENCES, NOT_ENOUGH_INLIERS, SINGULAR_H = 0, 1, 2, 3
BAD_ARGUMENT, DEVICE_ERROR, OUT_OF_MEMORY, CAPACITY_EXCEEDED, NO_DEVICE = -1, -2, -3, -4, -5
FINDER_PROJECTIVE, FINDER_NN, FINDER_DISTMAP, FINDER_KDTREE = 0, 1, 2, 3
ROBUST_NONE, ROBUST_CAUCHY = 0, 1


class Projector(C.Structure):
    _fields_ = [("canvas_cols", C.c_int32), ("angle_min", C.c_float), ("angle_max", C.c_float),
                ("range_min", C.c_float), ("range_max", C.c_float), ("col_offset", C.c_float)]


class SliceParams(C.Structure):
    _fields_ = [("finder", C.c_int32), ("projector", Projector), ("point_distance", C.c_float),
                ("normal_cos", C.c_float), ("max_distance", C.c_float), ("resolution", C.c_float),
                ("robustifier", C.c_int32), ("chi_threshold", C.c_float),
                ("min_num_correspondences", C.c_int32), ("sensor_in_robot", C.c_float * 3),
                ("kd_max_leaf_range", C.c_float), ("kd_min_leaf_points", C.c_int32)]


class AlignerParams(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("min_num_inliers", C.c_int32), ("damping", C.c_float),
                ("termination_chi_epsilon", C.c_float), ("enable_inlier_only_runs", C.c_int32),
                ("keep_only_inlier_correspondences", C.c_int32)]


class Prior(C.Structure):
    _fields_ = [("z", C.c_float * 3), ("omega", C.c_float * 9)]


class Correspondence(C.Structure):
    _fields_ = [("fixed_idx", C.c_int32), ("moving_idx", C.c_int32)]


class IterationStats(C.Structure):
    _fields_ = [("n_correspondences", C.c_int32), ("n_inliers", C.c_int32), ("n_outliers", C.c_int32),
                ("chi_inliers", C.c_float), ("chi_outliers", C.c_float), ("pair_digest_lo", C.c_uint32), ("pair_digest_hi", C.c_uint32)]

    @property
    def pair_digest(self) -> int:
        return (int(self.pair_digest_hi) << 32) | int(self.pair_digest_lo)


class Preprocessor(C.Structure):
    _fields_ = [("n_beams", C.c_int32), ("angle_min", C.c_float), ("angle_max", C.c_float), ("range_min", C.c_float),
                ("range_max", C.c_float), ("normal_point_distance", C.c_float), ("normal_min_points", C.c_int32),
                ("voxelize_resolution", C.c_float)]


class Batch(C.Structure):
    _fields_ = [("n_alignments", C.c_int32), ("n_slices", C.c_int32), ("slices", C.POINTER(SliceParams)),
                ("fixed", C.POINTER(C.c_void_p)), ("moving", C.POINTER(C.c_void_p)),
                ("fixed_index", C.POINTER(C.c_int32)), ("moving_index", C.POINTER(C.c_int32)),
                ("init_pose", C.POINTER(C.c_float)), ("prior", C.POINTER(Prior))]


# every symbol include/lsm2d.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("lsm2d_version", C.c_int, []),
    ("lsm2d_status_string", C.c_char_p, [C.c_int]),
    ("lsm2d_last_error", C.c_char_p, [_P]),
    ("lsm2d_create", C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    ("lsm2d_destroy", None, [_P]),
    ("lsm2d_synchronize", C.c_int, [_P]),
    ("lsm2d_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("lsm2d_get_option", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    ("lsm2d_last_kernel_ms", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("lsm2d_cloudset_create", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.POINTER(_P)]),
    ("lsm2d_cloudset_create_from_device", C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.POINTER(_P)]),
    ("lsm2d_cloudset_create_reserved", C.c_int, [_P, C.c_int64, C.POINTER(_P)]),
    ("lsm2d_cloudset_upload", C.c_int, [_P, _P, C.c_int64]),
    ("lsm2d_cloudset_download", C.c_int, [_P, C.c_int32, _P, C.c_int64, C.POINTER(C.c_int64)]),
    ("lsm2d_cloudset_destroy", None, [_P]),
    ("lsm2d_cloudset_num_clouds", C.c_int32, [_P]),
    ("lsm2d_cloudset_num_points", C.c_int64, [_P]),
    ("lsm2d_cloudset_cloud_size", C.c_int64, [_P, C.c_int32]),
    ("lsm2d_project", C.c_int, [_P, C.POINTER(Projector), _P, C.c_int32, _P, _P, _P, _P]),
    ("lsm2d_preprocess_scans", C.c_int, [_P, C.POINTER(Preprocessor), _P, C.c_int32, C.POINTER(_P)]),
    ("lsm2d_preprocess_scan_into", C.c_int, [_P, C.POINTER(Preprocessor), _P, _P]),
    ("lsm2d_clip_scene", C.c_int, [_P, C.POINTER(Projector), _P, C.c_int32, _P, _P, _P, C.POINTER(C.c_int32), _P]),
    ("lsm2d_cloudset_cloud_sizes", C.c_int32, [_P, _P, C.c_int32]),
    ("lsm2d_sweep_create", C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    ("lsm2d_sweep_destroy", None, [_P]),
    ("lsm2d_sweep_num_devices", C.c_int32, [_P]),
    ("lsm2d_sweep_last_error", C.c_char_p, [_P]),
    ("lsm2d_sweep_set_option", C.c_int, [_P, C.c_char_p, C.c_int64]),
    ("lsm2d_sweep_get_option", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    ("lsm2d_sweep_set_map", C.c_int, [_P, _P, C.c_int64]),
    ("lsm2d_sweep_set_scans", C.c_int, [_P, _P, _P, C.c_int32]),
    ("lsm2d_sweep_align", C.c_int, [_P, C.POINTER(AlignerParams), C.POINTER(SliceParams), C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    ("lsm2d_clip_scene_voxelized", C.c_int, [_P, C.POINTER(Projector), _P, C.c_int32, _P, _P, C.c_float, _P, C.POINTER(C.c_int32), _P]),
    ("lsm2d_merge_scene", C.c_int, [_P, C.POINTER(Projector), _P, _P, C.c_int32, _P, C.c_float, C.POINTER(C.c_int32), _P]),
    ("lsm2d_merge_scenes", C.c_int, [_P, C.POINTER(Projector), _P, C.c_int32, _P, _P, _P, C.c_float, C.POINTER(C.c_int32), _P]),
    ("lsm2d_find_correspondences", C.c_int,
     [_P, C.POINTER(SliceParams), _P, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    ("lsm2d_linearize", C.c_int,
     [_P, C.POINTER(SliceParams), _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, C.POINTER(IterationStats)]),
    ("lsm2d_align_batch", C.c_int, [_P, C.POINTER(AlignerParams), C.POINTER(Batch), _P, _P, _P, _P, _P]),
    ("lsm2d_align_batch_pairs", C.c_int, [_P, C.POINTER(AlignerParams), C.POINTER(Batch), _P, _P, _P, _P, _P, _P, C.c_int32, _P]),
    ("lsm2d_stats_capacity", C.c_int32, [C.POINTER(AlignerParams)]),
    ("lsm2d_pair_hash", C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    ("lsm2d_estimate_work", C.c_int, [_P, C.POINTER(Batch), _P]),
    ("lsm2d_align_batch_begin", C.c_int, [_P, C.POINTER(AlignerParams), C.POINTER(Batch), C.c_int32, C.POINTER(_P)]),
    ("lsm2d_align_batch_wait", C.c_int, [_P, _P, _P, _P, _P, _P]),
    ("lsm2d_preprocess_scans_refill", C.c_int, [_P, C.POINTER(Preprocessor), _P, C.c_int32, _P]),
]

_lib = None


def library_path() -> str:
    """liblsm2d_hip.so -- or, with LSM2D_EXPERIMENTS=1 in the environment, the experiments build of the same sources (tests / tuning only)."""
    return _build.lib_path()


def _preload_torch_hip_runtime():
    """PyTorch-ROCm bundles its own libamdhip64.so.7; /opt/rocm has another with the SAME soname.  A
    process must not end up with the system runtime resolved first and torch initialising on top of it
    ("No HIP GPUs are available"), so when torch is installed its runtime is loaded before ours."""
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.origin:
        p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(p):
            try:
                C.CDLL(p, mode=C.RTLD_GLOBAL)
            except OSError:
                pass


def load(build_if_missing: bool = True):
    """dlopen liblsm2d_hip.so and bind every declared symbol; raises if the library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path) or (build_if_missing and _build.is_stale()):
        if not build_if_missing:
            raise FileNotFoundError(path)
        _build.build()
    _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class Lsm2dError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        name = load().lsm2d_status_string(code).decode()
        super().__init__(f"{where}: {name} ({code}) {detail}".strip())


def check(code: int, where: str, ctx=None):
    if code < 0:
        detail = load().lsm2d_last_error(ctx).decode() if True else ""
        raise Lsm2dError(code, where, detail)
    return code
