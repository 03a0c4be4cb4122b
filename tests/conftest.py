import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def po():
    """The CPU oracle (test infrastructure)."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def small_workload():
    from srrg2_laser_slam_2d_amd import synth
    return synth.make_workload(6, 20000, seed=3)


@pytest.fixture(scope="session")
def ctx():
    """One lsm2d context on cuda:0 -- fails loudly if the HIP library or the GPU is missing."""
    from srrg2_laser_slam_2d_amd import api
    c = api.Context(0)
    yield c
    c.close()


# The A/B knobs of measured-and-rejected (or always-on) alternatives exist only in the library's experiments build (-DLSM2D_EXPERIMENTS,
# LSM2D_EXPERIMENTS=1 in the environment selects it: srrg2_laser_slam_2d_amd/build.py); value = what the shipped library does.
EXPERIMENT_DEFAULTS = {"cull_est_um": 0, "cull_est_urad": 40000, "results_to_host": 1, "two_stage": 0, "balance_notes": 1, "cull_keep": 1, "nn_qcache": 1,
                       "nn_lds_only": 1, "kd_modes": 1, "proj_modes": 1, "cull_block": 0, "kd_chain": 1, "grid_big_cells_x10": 50, "kd_scan_max_clouds": 8,
                       "kd_wide_min_points": 1024, "kd_wg_max_points": 16384, "estimate_reuse": 1, "xcd_lockstep": 0, "lane_streams": 1, "order_cluster": 0}


def has_experiments(ctx) -> bool:
    return ctx.get_option("experiments") == 1


def xset(ctx, **opts) -> bool:
    """Apply context options.  Public keys are set as they are.  An experiments key (or "cull" 2) is set when the library is the experiments build; the
    shipped library does not know it: asking for the value it has anyway is a no-op, asking for another one returns False -- the caller skips that
    variant (it runs when the suite is run against the experiments build)."""
    ok, x = True, has_experiments(ctx)
    for k, v in opts.items():
        if k in EXPERIMENT_DEFAULTS or (k == "cull" and v == 2):
            if x:
                ctx.set_option(k, v)
            elif v != EXPERIMENT_DEFAULTS.get(k):
                ok = False
        else:
            ctx.set_option(k, v)
    return ok


def need_experiments(ctx):
    if not has_experiments(ctx):
        pytest.skip("needs the experiments build of the library (LSM2D_EXPERIMENTS=1): this alternative is not in the shipped liblsm2d_hip.so")


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)
