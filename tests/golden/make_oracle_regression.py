"""Generates tests/golden/oracle_regression.json: ORACLE-GENERATED regression vectors (NOT reference outputs -- the
reference cannot be built here and holds no golden vector for this path).  They freeze what the restated
algorithm returns on small seeded inputs so that an accidental change of the oracle shows up on the CPU box.

    python tests/golden/make_oracle_regression.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po                      # noqa: E402
from srrg2_laser_slam_2d_amd import synth              # noqa: E402


def cases():
    wl = synth.make_workload(3, 8000, seed=42, n_beams=361)
    for i in range(3):
        yield i, wl, wl.scan_points[wl.scan_offsets[i]:wl.scan_offsets[i + 1]]


def main():
    out = {"note": "oracle-generated regression vectors, not reference outputs", "workload": "make_workload(3, 8000, seed=42, n_beams=361)", "cases": []}
    for i, wl, scan in cases():
        c = {"index": i}
        for name, sp in (("projective", po.slice_params(canvas_cols=361)), ("nn", po.slice_params(finder=po.FINDER_NN, max_distance=0.3)),
                         ("distmap", po.slice_params(finder=po.FINDER_DISTMAP, max_distance=0.5, resolution=0.1))):
            pairs = po.find(sp, scan, wl.map_points, wl.x0[i])
            r = po.align(po.aligner_params(10), [sp], [scan], [wl.map_points], wl.x0[i].astype(np.float64), double=True)
            # the fp32 mirror is a fixed sequence of IEEE operations (no libm since the fixed sin / cos / log): its results are the same
            # BITS on every host, in both summation orders
            f32 = {}
            for tag, dev in (("sequential", False), ("device_order", True)):
                rf = po.align(po.aligner_params(10, device_order=dev), [sp], [scan], [wl.map_points], wl.x0[i])
                f32[tag] = {"pose_hex": [float(v).hex() for v in rf["pose"]], "status": int(rf["status"]),
                            "n_corr": [int(st.n_corr) for st in rf["stats"]], "chi_in_hex": [float(st.chi_in).hex() for st in rf["stats"]]}
            c[name + "_fp32"] = f32
            c[name] = {"n_pairs": int(len(pairs)), "pairs_checksum": int((pairs.astype(np.int64) * np.array([1000003, 7919])).sum() % (2 ** 31)),
                       "pose_after_10_its_fp64": [float(v) for v in r["pose"]], "status": int(r["status"]),
                       "n_corr_first": int(r["stats"][0].n_corr), "n_corr_last": int(r["stats"][-1].n_corr), "chi_first": float(r["stats"][0].chi_in)}
        out["cases"].append(c)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
