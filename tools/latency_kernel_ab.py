#!/usr/bin/env python3
"""A/B of the two one-workgroup-per-alignment kernels on single-slice projective calls too small to fill the chip:
k_align (align_path 1) against the latency kernel k_align_pair (align_path 3).  Prints kernel and wall time per call and checks that
the results are bit-identical.      python tools/latency_kernel_ab.py"""
import json, math, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import api, synth


def main():
    ctx = api.Context(0)
    world = synth.make_world(3)
    out = []
    for name, n, n_map, its, beams, prior in (("configs[0]: 1 scan vs 10k map, 20 its", 1, 10000, 20, 1081, False),
                                              ("single-laser tracker: 1 scan vs 700-point scene, 10 its, prior", 1, 700, 10, 721, True),
                                              ("16 candidates vs 10k map", 16, 10000, 20, 1081, False),
                                              ("64 candidates vs 30k map", 64, 30000, 20, 1081, False),
                                              ("256 candidates vs 10k map", 256, 10000, 20, 1081, False)):
        wl = synth.make_workload(n, n_map, seed=5, n_beams=beams, world=world)
        fixed = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets); moving = api.CloudSet(ctx, wl.map_points)
        al = api.MultiAligner2D(ctx, max_iterations=its, min_num_inliers=10)
        al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(
            api.CorrespondenceFinderProjective2f(ctx, api.PointNormal2fProjectorPolar(beams, -math.pi, math.pi, 0.3, 30.0)), min_num_correspondences=10))
        pri = [(wl.x0[i], np.eye(3, dtype=np.float32) * 50.0) for i in range(n)] if prior else None
        rec = {"case": name}
        res = {}
        for path in (1, 3):
            ctx.set_option("align_path", path)
            for _ in range(200):
                r = al.compute_batch([fixed], [moving], wl.x0, priors=pri)
            k = []; t0 = time.perf_counter()
            for _ in range(500):
                r = al.compute_batch([fixed], [moving], wl.x0, priors=pri); k.append(r.kernel_ms)
            rec["path%d" % path] = {"kernel_ms": float(np.mean(k)), "wall_ms": 1e3 * (time.perf_counter() - t0) / 500, "ran": ctx.get_option("last_align_path")}
            res[path] = r
        ctx.set_option("align_path", 0)
        rec["bit_identical"] = bool(np.array_equal(res[1].pose, res[3].pose) and np.array_equal(res[1].information, res[3].information) and np.array_equal(res[1].status, res[3].status))
        rec["status_ok"] = bool(np.all(res[1].status == 0))
        out.append(rec); print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
