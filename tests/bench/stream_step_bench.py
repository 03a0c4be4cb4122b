#!/usr/bin/env python3
"""The streamed pipeline through the C ABI alone: builds tests/cpp/stream_step_bench.cpp with g++ (+ the HIP runtime's host API for pinned memory) against
liblsm2d_hip.so, feeds it `--batches` distinct batches of raw range vectors of the workload bench.py --stream uses (1000 scans x 1081 beams vs one 100 000-point
map, 20 iterations, role A, projective finder) and prints its JSON line -- what a C++ host of the reference's kind sustains from ranges to poses, every step
bit-identical to the synchronous calls, beside bench.py --stream's figure for the Python host.
    python tests/bench/stream_step_bench.py [--steps 300] [--scans 1000] [--map-points 100000]"""
import argparse, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import synth


def build(exe):
    lib = os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "lib")
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "srrg2_laser_slam_2d_amd", "host"),
                    os.path.join(ROOT, "tests", "cpp", "stream_step_bench.cpp"), "-L" + lib, "-llsm2d_hip", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)


def run(steps=300, warmup=300, scans=1000, map_points=100000, iterations=20, beams=1081, batches=4, seed=0, ahead=1, workdir=None):
    world = synth.make_world(seed)
    m = synth.make_map(world, map_points, seed=seed)
    a0, a1 = -0.75 * np.pi, 0.75 * np.pi
    rg, x0 = [], []
    for k in range(batches):
        poses = synth.sample_poses(world, scans, seed=seed + 7919 * (k + 1))
        rg.append(synth.make_scan_ranges(world, poses, n_beams=beams, angle_min=a0, angle_max=a1, seed=seed + k))
        x0.append(synth.initial_guesses(poses, seed=seed + k)[1].astype(np.float32))
    import contextlib
    if workdir:
        os.makedirs(workdir, exist_ok=True)
    with (contextlib.nullcontext(workdir) if workdir else tempfile.TemporaryDirectory()) as d:
        exe = os.path.join(d, "stream_step_bench")
        build(exe)
        np.ascontiguousarray(m, np.float32).tofile(os.path.join(d, "map.bin"))
        np.ascontiguousarray(np.stack(rg), np.float32).tofile(os.path.join(d, "ranges.bin"))
        np.ascontiguousarray(np.stack(x0), np.float32).tofile(os.path.join(d, "x0.bin"))
        cmd = [exe, os.path.join(d, "map.bin"), os.path.join(d, "ranges.bin"), os.path.join(d, "x0.bin"), str(scans), str(beams), str(batches),
               str(steps), str(warmup), str(iterations), repr(float(np.float32(a0))), repr(float(np.float32(a1))), str(ahead)]
        if workdir:      # (kept: the driver, its inputs and the command line -- what a rocprofv3 kernel trace of the C++ host runs)
            with open(os.path.join(d, "cmd.txt"), "w") as f:
                f.write(" ".join(cmd) + "\n")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError("stream_step_bench failed (%d): %s | %s" % (r.returncode, r.stderr[-2000:], r.stdout[-1000:]))
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["workload"] = "STREAM through the bare C ABI: every step %d new %d-beam range vectors (pinned) -> preprocessed on the device -> aligned vs one %d-point map, %d iterations, one step in flight" % (scans, beams, map_points, iterations)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300); ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--scans", type=int, default=1000); ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20); ap.add_argument("--beams", type=int, default=1081); ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--workdir", default=None, help="keep the built driver, its inputs and cmd.txt (its command line) here")
    ap.add_argument("--ahead", type=int, default=1, help="1: three scan sets, the next step's scans refilled behind this step's begin; 0: two sets, refill just before begin")
    a = ap.parse_args()
    print(json.dumps(run(a.steps, a.warmup, a.scans, a.map_points, a.iterations, a.beams, a.batches, ahead=a.ahead, workdir=a.workdir)))
