"""Build liblsm2d_hip.so (the C ABI of include/lsm2d.h) for gfx950 with hipcc, in-tree.

    python -m srrg2_laser_slam_2d_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "liblsm2d_hip.so")
# The same sources with -DLSM2D_EXPERIMENTS: the measured-and-rejected alternatives of DESIGN App. A (second launch form, row-major culled stream, the A/B
# option keys) compiled in.  NOT the product: the GPU suite run with LSM2D_EXPERIMENTS=1 in the environment (its variant tests are skipped otherwise) and tuning scripts load it.
LIB_PATH_EXPERIMENTS = os.path.join(LIB_DIR, "liblsm2d_hip_experiments.so")
SOURCES = [os.path.join(CSRC, "lsm2d_capi.hip")]
# every header and include part under csrc/ (the kernels by family: lsm2d_k_*.h; the host side's parts: lsm2d_capi_*.inc) + the ABI header
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))) + [os.path.join(ROOT, "include", "lsm2d.h")]

# -ffp-contract=off: every fused multiply-add in the kernels is explicit, so column indices and
# z-buffer winners are reproducible bit-for-bit by an IEEE CPU (see csrc/lsm2d_device.h).
# -fno-slp-vectorize: the SLP pass pairs scalar fp32 operations into v_pk_*_f32, which issue slower than the two scalar
# instructions they replace on gfx950 (measured: DESIGN.md section 5).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def source_hash() -> str:
    """sha256 over the kernel sources, the ABI header and the compiler flags: what the counters under profiles/ were taken on
    (bench.py reports PMC-derived numbers only when this matches profiles/counters.json)."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(SOURCES + HEADERS):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    extra = os.environ.get("LSM2D_EXTRA_HIPCC_FLAGS", "").split()      # a variant build is another instruction stream: its counters are its own
    if extra:
        h.update(b"\0extra:" + " ".join(extra).encode())
    return h.hexdigest()


def experiments_selected() -> bool:
    return os.environ.get("LSM2D_EXPERIMENTS", "0") not in ("", "0")


def lib_path(experiments: bool | None = None) -> str:
    if experiments is None:
        experiments = experiments_selected()
    return LIB_PATH_EXPERIMENTS if experiments else LIB_PATH


def is_stale(experiments: bool | None = None) -> bool:
    path = lib_path(experiments)
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, experiments: bool | None = None) -> str:
    if experiments is None:
        experiments = experiments_selected()
    LIB_PATH = lib_path(experiments)
    if not force and not is_stale(experiments):
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("LSM2D_EXTRA_HIPCC_FLAGS", "").split()      # tuning experiments only
    if experiments:
        extra = ["-DLSM2D_EXPERIMENTS"] + extra
    # compile next to the target and rename: several ranks of one job may get here together (torchrun), and none of them must
    # ever dlopen a half-written file
    tmp = "%s.%d.tmp" % (LIB_PATH, os.getpid())
    cmd = [hipcc(), *HIPCC_FLAGS, *extra, "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", tmp, *SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments=True if "--experiments" in sys.argv else None))
