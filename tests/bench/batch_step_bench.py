#!/usr/bin/env python3
"""BASELINE configs[1] through the C ABI alone: builds tests/cpp/batch_step_bench.cpp with g++ against liblsm2d_hip.so, feeds it the same synthetic
batch bench.py uses (1000 scans x 1081 beams vs one 100 000-point map, 20 iterations, role A, projective finder) and prints its JSON line -- what a
C++ host of the reference's kind pays per batch, beside bench.py's figure for the Python host.
    python tests/bench/batch_step_bench.py [--steps 300] [--scans 1000] [--map-points 100000]"""
import argparse, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300); ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--scans", type=int, default=1000); ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20); ap.add_argument("--beams", type=int, default=1081)
    a = ap.parse_args()
    wl = synth.make_workload(a.scans, a.map_points, seed=0, n_beams=a.beams)
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "batch_step_bench"); lib = os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "lib")
        subprocess.run(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "batch_step_bench.cpp"),
                        "-L" + lib, "-llsm2d_hip", "-Wl,-rpath," + lib, "-o", exe], check=True)
        np.ascontiguousarray(wl.map_points, np.float32).tofile(os.path.join(d, "map.bin"))
        np.ascontiguousarray(wl.scan_points, np.float32).tofile(os.path.join(d, "scans.bin"))
        np.ascontiguousarray(wl.scan_offsets, np.int32).tofile(os.path.join(d, "offs.bin"))
        np.ascontiguousarray(wl.x0, np.float32).tofile(os.path.join(d, "x0.bin"))
        r = subprocess.run([exe, os.path.join(d, "map.bin"), os.path.join(d, "scans.bin"), os.path.join(d, "offs.bin"), os.path.join(d, "x0.bin"),
                            str(a.steps), str(a.warmup), str(a.iterations), str(a.beams)], check=True, capture_output=True, text=True, timeout=600)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    err = np.abs(np.float32(out["pose0"]) - wl.x_true[0].astype(np.float32))
    out["pose0_error_m_rad"] = [float(max(err[0], err[1])), float(err[2])]
    out["workload"] = "configs[1]: %d scans x %d beams vs one %d-point map, %d iterations, role A, projective" % (a.scans, a.beams, a.map_points, a.iterations)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
