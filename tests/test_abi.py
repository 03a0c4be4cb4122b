"""CPU checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every
symbol include/lsm2d.h declares.  No compute calls here (no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lsm2d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lsm2d_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from srrg2_laser_slam_2d_amd import _capi, build
    path = build.build()
    assert os.path.exists(path)
    lib = C.CDLL(path)
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), name
    bound = {s[0] for s in _capi.SYMBOLS}
    assert set(declared) == bound
    l = _capi.load()
    assert l.lsm2d_version() == 112
    assert l.lsm2d_status_string(0) == b"Success" and l.lsm2d_status_string(-4) == b"CapacityExceeded"


def test_no_silent_cpu_fallback():
    """Without a HIP device the product path must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from srrg2_laser_slam_2d_amd import api
    with pytest.raises(api.Lsm2dError) as ei:
        api.Context(0)
    assert ei.value.code == -5


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "srrg2_laser_slam_2d_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in src and "lsm2d_oracle" not in src and "lsmo_" not in src, f


def test_struct_layouts_match_header_sizes():
    from srrg2_laser_slam_2d_amd import _capi
    assert C.sizeof(_capi.Projector) == 24
    assert C.sizeof(_capi.SliceParams) == 4 + 24 + 16 + 4 + 4 + 4 + 12
    assert C.sizeof(_capi.AlignerParams) == 12
    assert C.sizeof(_capi.Prior) == 48
    assert C.sizeof(_capi.Correspondence) == 8
    assert C.sizeof(_capi.IterationStats) == 20
