"""Generates tests/golden/tracker_chain.json: ORACLE-GENERATED digests (NOT reference outputs -- the reference cannot be built here,
SURVEY.md 8c) of an 8-step live-tracker chain: raw ranges -> preprocessed scans -> clipped local map -> pose of the two-slice aligner
with odometry prior (fp32, the kernels' summation order) -> merged local map.  tests/test_oracle.py holds the oracle to it on the CPU
box, tests/test_gpu_configs_at_size.py the HIP path on the MI355X (no oracle call there).

    python tests/golden/make_tracker_chain.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po          # noqa: E402
import tracker_chain                       # noqa: E402

if __name__ == "__main__":
    out = {"note": "oracle-generated digests (sha256[:20] of the float32 arrays), not reference outputs", "scenario": "tests/tracker_chain.py scenario(8)",
           "steps": tracker_chain.run_oracle(po, 8)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tracker_chain.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "final map", out["steps"][-1]["map_points"], "points")
    # BASELINE configs[2] at its stated size: 1000 steps of the same chain (MULTI parameters: 721 columns, 10 iterations, two WithSensor
    # laser slices + odometry prior, clip + merge), digests of every 50th step (apps/visual_test_tracker_2d.cpp:167-183 is the usage contract)
    out = {"note": "oracle-generated digests (sha256[:20] of the float32 arrays), not reference outputs", "scenario": "tests/tracker_chain.py scenario(1000)",
           "steps_total": 1000, "record_every": 50, "steps": tracker_chain.run_oracle(po, 1000, record_every=50)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tracker_replay_1000.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, "final map", out["steps"][-1]["map_points"], "points")
