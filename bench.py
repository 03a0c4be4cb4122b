#!/usr/bin/env python3
"""bench.py -- scan-to-map alignments/sec on MI355X (BASELINE.json metric), one process per GPU.

Workload (BASELINE.json configs[1]): a batch of 1000 synthetic 1081-beam scans against ONE 100k-point
local map, 20 Gauss-Newton iterations per alignment, reference role assignment (fixed = scan,
moving = map, projective finder -- SURVEY.md section 8d "role A / projective").  One step = one pass of the
batch through lsm2d_align_batch with the clouds already resident in HBM and NEW START POSES every step
(--pose-sets sets in rotation: the 12 KB upload and the placement's estimate are inside the step; the
step with identical poses every time, which the library answers from what it kept, is the labelled
`same_poses_every_step` block).  With N GPUs every rank aligns its own 1000 scans (weak scaling) against
the map broadcast from rank 0 over RCCL; there is no data-path collective.
Blocks beside the headline on the default N=1 line: same_poses_every_step, sum_order_1 (the reference's
order of summation: bitwise the sequential fp32 oracle), pipelined (two batches in flight), streamed
(ranges in, poses out), also (configs[4], configs[3]), cpu_baseline (+ all_cores).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

The roofline block: the map (0.8 MB of coordinates at 1e5 points) is served from every XCD's L2, so
the bound that binds is VALU ISSUE, not HBM.  roofline.achieved / peak are wave64 VALU issue slots per second:
  achieved = (SQ_INSTS_VALU + transcendentals, which take two slots) per launch / launch time (HIP events, this run)
  peak     = 1024 SIMDs x the clock the chip held INSIDE this run's launches (s_memtime / s_memrealtime stamps) / 2 cycles
SQ_INSTS_VALU and the HBM traffic come from the rocprofv3 --pmc passes recorded in profiles/counters.json; they are only
reported when the sha256 of the kernel sources they were taken on matches the library that ran (else null + a warning).
The SURVEY 8(d) algorithmic-bytes figure is kept as hbm.effective_l2_served_GBs next to the real DRAM rate.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
N_SIMD = 256 * 4             # same guide: 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles,
SPEC_CLOCK_MHZ = 2400.0      # same guide: max clock (spec)
L2_PEAK_GBS = 34500.0        # same guide: ~34.5 TB/s aggregate over the eight per-XCD L2s
VALU_CYCLES = 2.0            # a transcendental (the stream's one v_rsq_f32 per point) over 4: one slot more than SQ_INSTS_VALU counts for it
L3_BYTES = 256 * 2 ** 20     # same guide: Infinity Cache (L3), die-level
L3_GATHER_PEAK_GBS = 8600.0  # same guide, "Indexed rows": a cached random gather out of the Infinity Cache reads at 8.6 TB/s chip-wide
FP32_VECTOR_PEAK_TFLOPS = 157.3   # same guide: fp32 vector (non-matrix) peak
FLOPS_PER_POINT = 60.0       # "A ~ 60 N_m" algorithmic flops per projected point


def algorithmic_bytes_per_alignment(role: str, finder: str, n_map: int, n_scan_mean: float, bins: int, iterations: int) -> float:
    """SURVEY.md section 8(d) table, per GN iteration (+ one-off term):
    A/projective  16*N_m (stream moving) + 48*Bins (canvas key write+read, fixed-cell read, winner gather) + 64; once 16*N_s
    A/nn          16*N_m + 28*C (C counted at N_m) + 64; once 16*N_s
    B/nn          16*N_s + N_s*(24*d + 8*L + 8) + 64 with L = 20, d = ceil(log2(N_m/L))"""
    if finder == "projective":      # the same row with the clouds' roles swapped (role B: the scan is streamed every iteration, the map projected once)
        n_moving, n_fixed = (n_map, n_scan_mean) if role == "A" else (n_scan_mean, n_map)
        return iterations * (16.0 * n_moving + 48.0 * bins + 64.0) + 16.0 * n_fixed
    if role == "A" and finder == "nn":
        return iterations * (16.0 * n_map + 28.0 * n_map + 64.0) + 16.0 * n_scan_mean
    if finder in ("nn", "kdtree"):      # SURVEY 8(d) row "B / NN": a descent of d nodes of 24 B + a leaf of L points of 8 B + the matched fixed point, per query
        nq, nf = (n_scan_mean, n_map) if role == "B" else (n_map, n_scan_mean)
        d = max(1, math.ceil(math.log2(max(nf, 40.0) / 20.0)))
        return iterations * (16.0 * nq + nq * (24.0 * d + 8.0 * 20 + 8.0) + 64.0)
    if finder == "distmap":      # per query: its own 16 B, one 4-byte parent lookup instead of a tree walk, 16 B of the parent
        nq = n_map if role == "A" else n_scan_mean
        return iterations * (36.0 * nq + 64.0)
    raise SystemExit("unsupported role/finder combination")


def host_cpu() -> dict:
    """Model string and PHYSICAL core count of the host (BASELINE.md section 3 asks for both next to the CPU number)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip(); v = v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core)); phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    # the cgroup's CPU quota (cpu.max: "<quota us> <period us>" or "max ..."): a 256-thread affinity mask with a 16-CPU quota runs 16 threads' worth, whatever is started
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota = None if txt[0] == "max" else float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0]); quota = None if q <= 0 else q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"model": model, "physical_cores": len(cores) or None, "threads_allowed": len(os.sched_getaffinity(0)), "cgroup_cpu_quota": quota}


def rank_affinity(allowed, local_rank: int, world: int):
    """The cores rank `local_rank` of `world` pins itself to: a contiguous slice of the cores the process may use (the remainder goes to the first ranks); everybody
    keeps the whole set when there are fewer cores than ranks.  (A step is a 0.8 ms launch loop: a rank whose host thread migrates between sockets shows it.)"""
    allowed = sorted(allowed)
    if world <= 1 or len(allowed) < world:
        return allowed
    per, extra = divmod(len(allowed), world)
    lo = local_rank * per + min(local_rank, extra)
    return allowed[lo: lo + per + (1 if local_rank < extra else 0)]


_RANK_NOTE = {"path": None, "data": {}}


def rank_note(stage: str, **kw) -> None:
    """Every rank of an N > 1 run keeps a JSON file of its own up to date -- gpurun_out/bench_rank<r>.json, or under LSM2D_BENCH_RANK_DIR -- with the stage it has
    reached (init / inputs / warmup / timed / done) and what it measured: a rank that hangs or dies is visible from outside by the stage its file stopped at."""
    if not _RANK_NOTE["path"]:
        return
    _RANK_NOTE["data"].update(kw, stage=stage, t=time.time())
    tmp = _RANK_NOTE["path"] + ".tmp"
    try:
        with open(tmp, "w") as f:
            json.dump(_RANK_NOTE["data"], f)
        os.replace(tmp, _RANK_NOTE["path"])
    except OSError:
        pass


def load_counters(cfg_key: str):
    """profiles/counters.json (written by tools/write_counters.py from the rocprofv3 --pmc passes) if it was taken on the
    kernels that are running now, else None."""
    from srrg2_laser_slam_2d_amd import build as hip_build
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        cj = json.load(open(path))
    except (OSError, ValueError):
        return None, "profiles/counters.json missing"
    if cj.get("csrc_sha256") != hip_build.source_hash():
        return None, "profiles/counters.json was taken on other kernel sources (sha256 mismatch): PMC-derived fields are null"
    ent = cj.get("configs", {}).get(cfg_key)
    if not ent:
        return None, "profiles/counters.json has no entry for this workload (%s)" % cfg_key
    ent = dict(ent); ent["source"] = cj.get("source")
    return ent, None


def working_set_bytes(role: str, finder: str, scans: int, map_points: int, n_scan_mean: float) -> float:
    """What one launch of the projective role-A batch touches: the map's lane-chunked copy (8 B per point), its AoS rows (16 B per point: one gather per z-buffer
    winner), chunk / block circles (65 KB), the scans' AoS rows.  Against the 256 MiB Infinity Cache this says whether L2 misses can be served on-die."""
    if finder == "projective" and role == "A":
        return 8.0 * map_points + 16.0 * map_points + 8 * 512 * 16 + 16.0 * scans * n_scan_mean
    return 32.0 * map_points + 32.0 * scans * n_scan_mean


def build_roofline(role, finder, scans, map_points, iterations, beams, cauchy, n_unique, n_scan_mean, k_ms, k_samples, clk, wg_ms, use_counters=True):
    """The roofline block of one measured workload (the headline line and every entry of `also`).  use_counters False: a workload no PMC pass was taken on (the
    streamed pipeline's preprocessed scans): launch time, clock and the algorithmic figure only -- the instruction counts of OTHER clouds would price it wrongly."""
    cfg_key = "role%s/%s/scans%d/map%d/it%d/beams%d" % (role, finder, scans, map_points, iterations, beams)
    if cauchy > 0:                       # a robustified run is another instruction stream (the log, the weights): its own counters
        cfg_key += "/cauchy%g" % cauchy
    if n_unique != scans:                # scans shared through the index array: another memory pattern
        cfg_key += "/unique%d" % n_unique
    bytes_per_alignment = algorithmic_bytes_per_alignment(role, finder, map_points, n_scan_mean, beams, iterations)
    counters, warn = load_counters(cfg_key) if use_counters else (None, None)
    effective = bytes_per_alignment * scans / (k_ms * 1e-3) / 1e9
    ws = working_set_bytes(role, finder, n_unique if 0 < n_unique < scans else scans, map_points, n_scan_mean)      # (candidates that share scans through an index array: the distinct scans)
    roof = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G wave64-VALU issue slots/s", "frac": None, "traffic": None,
            "kernel": "k_align", "kernel_ms": k_ms, "kernel_ms_samples": k_samples, "clock_mhz_in_kernel": clk, "workgroup_lifetime_ms": wg_ms,
            "hbm": {"effective_l2_served_GBs": effective, "effective_over_hbm_peak": effective / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": bytes_per_alignment * scans, "peak_GBs": HBM_PEAK_GBS},
            # Round 5: what the memory-side counters of this pool CAN tell.  FETCH_SIZE / TCC_EA0_RDREQ count the L2s' MISSES going out on the fabric; whether the
            # Infinity Cache (256 MiB) or the DRAM behind it served them they do not say -- TCC_EA0_RDREQ_DRAM counts the same requests to four digits at 1M, 4M
            # and 16M map points (profiles/r05/pmc_k_align_map*_r05a.csv): it names the address space, not the level that answered.  So the line carries the fabric
            # rate, says whether the launch's working set fits the Infinity Cache, and claims NO DRAM figure.
            "fabric_GBs": None, "dram_GBs": None, "working_set_MB": ws / 1e6, "l3_resident": bool(ws < 0.9 * L3_BYTES),
            "note": "frac = VALU issue slots used / slots the 1024 SIMDs have at the 2.4 GHz spec clock (frac_at_in_kernel_clock: at the clock measured inside the "
                    "launch).  fabric_GBs = the L2s' miss traffic (2 x FETCH_SIZE + WRITE_SIZE per launch / launch time): Infinity Cache or DRAM, the counters cannot "
                    "tell which (dram_GBs stays null); l3_resident says whether the working set fits the 256 MiB Infinity Cache.  hbm.* keeps the SURVEY 8(d) algorithmic figure"}
    # what the L2s serve: the kernel's vector-memory read instructions x 1 KiB (16 bytes per lane) against the 34.5 TB/s of the eight L2s
    # (MI355X_MICROARCH.md, "L2 (per XCD)"); the instruction count is a committed counter like the VALU count (profiles/counters.json)
    if counters and counters.get("vmem_rd_insts_per_launch"):
        l2_bytes = counters["vmem_rd_insts_per_launch"] * 1024.0
        roof["l2_served"] = {"bytes_per_launch": l2_bytes, "GBs": l2_bytes / (k_ms * 1e-3) / 1e9, "peak_GBs": L2_PEAK_GBS,
                             "frac": l2_bytes / (k_ms * 1e-3) / 1e9 / L2_PEAK_GBS}
    if clk:
        roof["peak"] = N_SIMD * clk * 1e6 / VALU_CYCLES / 1e9
    if counters and clk:
        # nominal costs (MI355X_MICROARCH.md): a wave64 VALU instruction issues over 2 cycles, a transcendental (one v_rsq_f32 per point
        # slot of the stream) over 4, i.e. one slot more than SQ_INSTS_VALU counts for it.  (Measured, tools/valu_issue_probe.hip: 2.15 and
        # ~12-18 cycles -- the stream_floor block below prices the launch with the measured costs instead.)
        slots = counters["valu_insts_per_launch"] + counters.get("trans_insts_per_launch", 0.0)
        roof["achieved"] = slots / (k_ms * 1e-3) / 1e9
        # `frac` is priced against the SPEC clock (2.4 GHz) since round 4 -- the peak the guide prints; the chip holds 2.1-2.3 GHz under this load,
        # so the figure against the clock measured inside the launch (the slots the SIMDs really had) reads ~7 % higher: kept beside it
        roof["peak_at_in_kernel_clock"] = roof["peak"]
        roof["peak"] = N_SIMD * SPEC_CLOCK_MHZ * 1e6 / VALU_CYCLES / 1e9
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["frac_at_spec_clock"] = roof["frac"]
        roof["frac_at_in_kernel_clock"] = roof["achieved"] / roof["peak_at_in_kernel_clock"]
        if counters.get("wave_points_per_launch"):
            # what of that issue rate is the reference's arithmetic: SURVEY 8(d)'s 60 flop per projected point x the points the kernel really visits
            # (counted: one v_rsq_f32 wave-instruction per 64 point visits) against the fp32 vector peak
            uf = FLOPS_PER_POINT * counters["wave_points_per_launch"] * 64.0 / (k_ms * 1e-3) / 1e12
            roof["useful_flops"] = {"TFLOPs": uf, "peak_TFLOPs": FP32_VECTOR_PEAK_TFLOPS, "frac": uf / FP32_VECTOR_PEAK_TFLOPS, "flops_per_point_visit": FLOPS_PER_POINT}
        roof["counters_are"] = "SQ_INSTS_VALU / vmem / fabric bytes per launch are COMMITTED constants (profiles/counters.json, checked against a hash of the kernel sources); only the launch time and the clock are measured live"
        roof["valu_insts_per_launch"] = counters["valu_insts_per_launch"]
        roof["trans_insts_per_launch"] = counters.get("trans_insts_per_launch")
        roof["traffic"] = counters.get("hbm_bytes_per_launch")      # (the key's name is round 1's; what it holds is the FABRIC traffic: see the note)
        if roof["traffic"]:
            roof["fabric_GBs"] = roof["traffic"] / (k_ms * 1e-3) / 1e9
            # which level that traffic could be bound by: an on-die gather (8.6 TB/s measured in the guide) while the working set fits the Infinity Cache, the HBM peak beyond
            fabric_peak = L3_GATHER_PEAK_GBS if roof["l3_resident"] else HBM_PEAK_GBS
            roof["fabric_frac"] = roof["fabric_GBs"] / fabric_peak
            if roof["fabric_frac"] > roof["frac_at_in_kernel_clock"]:
                # a mode whose fabric traffic is the larger fraction of its ceiling than its VALU issue is AT THE CLOCK THE CHIP HELD (a distance map per scan:
                # 3 GB of maps gathered at random; the projective stream over a 4M-point map and beyond: L2 hit rate 17 % and falling)
                roof["valu_issue"] = {"achieved": roof["achieved"], "peak": roof["peak"], "frac": roof["frac"], "frac_at_in_kernel_clock": roof["frac_at_in_kernel_clock"], "unit": roof["unit"]}
                roof.update(bound="fabric (Infinity Cache)" if roof["l3_resident"] else "fabric (Infinity Cache + HBM)", achieved=roof["fabric_GBs"], peak=fabric_peak, unit="GB/s",
                            frac=roof["fabric_frac"])
        if counters.get("point_visits_frac") is not None:
            # the exact culling against the fixed canvas (round 3): which fraction of the (point, iteration) visits of the plain stream the
            # kernel still makes -- counted (v_rsq_f32 wave-instructions), not modelled; the results are bit-identical either way
            roof["culling"] = {"point_visits_frac": counters["point_visits_frac"], "wave_point_visits_per_launch": counters.get("wave_points_per_launch"),
                               "wave_point_visits_without_culling": counters.get("wave_points_full")}
        roof["counters_source"] = counters.get("source")
        if counters.get("stream_cycles_per_wave_point") and counters.get("wave_points_per_launch"):
            # the second, sharper yardstick: what the 1024 SIMDs need for THIS instruction stream when nothing else is in the way
            # (tools/valu_issue_probe.hip runs csrc's project_point_stream on register-resident points: cycles per point of a wave)
            cyc = counters["stream_cycles_per_wave_point"] * counters["wave_points_per_launch"] / N_SIMD
            floor_ms = cyc / (clk * 1e6) * 1e3
            roof["stream_floor"] = {"cycles_per_wave_point": counters["stream_cycles_per_wave_point"], "floor_ms_at_measured_clock": floor_ms,
                                    "frac": floor_ms / k_ms,
                                    "note": "kernel time / (point visits x the stream's own measured issue cost): bin walk, reductions, 3x3 solves and "
                                            "barriers of the other workgroups on a CU run underneath the stream when this is ~1 (the probe is a launch of its own: "
                                            "+-2 % between passes, so values just above 1 are its error, not a faster-than-floor kernel)"}
    if warn:
        roof["warning"] = warn
        print("bench.py: " + warn, file=sys.stderr)
    return roof


def measure_pipelined(ctx, prep_a, prep_b, x0_sets, want_sets, args, roof) -> dict:
    """The SAME resident-input step with two batches in flight (lsm2d_align_batch_begin / _wait: begin(k) ; wait(k - 1), two prepared batches alternating): each
    asynchronously begun batch launches on its lane's own stream, so the younger launch's workgroups fill the slots the older one's tail leaves free (the mean
    workgroup ends 10 % before its launch).  Every run's results are compared bit for bit with the synchronous step's.  The headline `value` stays the synchronous
    step (one launch at a time: the roofline block's launch duration is that of a launch alone); this block is what a host with a queue of batches gets."""
    n_sets = len(x0_sets)
    pair = (prep_a, prep_b)
    steps = max(250, min(args.steps, 2000)); warm = 30      # (its own length: the driver's 20 timed steps are too few for a pipeline to settle; ~0.2 s)
    kt = ctx.get_option("kernel_timing"); ctx.set_option("kernel_timing", 0)      # (events around two overlapping launches time nothing meaningful)
    bad = 0
    try:
        t0 = None
        for k in range(warm + steps):
            if k == warm:
                t0 = time.perf_counter()
            if n_sets > 1:
                pair[k & 1].set_init_poses(x0_sets[k % n_sets])      # new start poses every batch: upload and placement inside the step
            pair[k & 1].begin()
            if k:
                r = pair[(k - 1) & 1].wait(); w = want_sets[(k - 1) % n_sets]
                bad += 0 if (np.array_equal(r.pose, w.pose) and np.array_equal(r.status, w.status)) else 1
        r = pair[(warm + steps - 1) & 1].wait(); w = want_sets[(warm + steps - 1) % n_sets]
        elapsed = time.perf_counter() - t0
        bad += 0 if (np.array_equal(r.pose, w.pose) and np.array_equal(r.status, w.status)) else 1
    finally:
        ctx.set_option("kernel_timing", kt)
    n = int(want_sets[0].pose.shape[0]); ms = elapsed / steps * 1e3
    out = {"value": n * steps / elapsed, "unit": "alignments/s", "ms_per_step": ms, "steps": steps, "batches_in_flight": 2,
           "runs_that_differed_from_the_synchronous_step": bad, "parity_ok": bad == 0,
           "pose_sets": n_sets,
           "note": "begin(k) ; wait(k - 1) over the resident scans, new start poses every batch (%d sets in rotation): launches of two lanes overlap on two streams; results bitwise those of the synchronous step" % n_sets}
    v, t = roof.get("valu_insts_per_launch"), roof.get("trans_insts_per_launch")
    if v and roof.get("peak"):      # the chip's VALU issue rate over the whole region: committed counters x launches / wall time, against the same peak as roofline.frac
        out["valu_issue_frac_at_spec_clock"] = (v + (t or 0.0)) / (ms * 1e-3) / 1e9 / roof["peak"]
    return out


def measure_also(ctx, api, synth, world_geom, wl, scan_set, args, x0_sets) -> list:
    """The other single-GPU BASELINE configurations, timed ONCE inside the default run so that the driver's record holds them: each entry has
    its own parity gate (noise-free data: the generating pose within 1e-4 m / 1e-4 rad), wall ms per step, kernel ms by HIP events, the in-kernel clock and a
    roofline block built like the headline's."""
    import time as _t
    entries = []

    def timed(prepared, warm, steps, sets):
        # (new start poses every step, as in the headline: `sets` in rotation)
        for i in range(warm):
            prepared.set_init_poses(sets[i % len(sets)]); prepared.run()
        k, c, w = [], [], []
        t0 = _t.perf_counter()
        for i in range(steps):
            prepared.set_init_poses(sets[(warm + i) % len(sets)])
            r = prepared.run(); k.append(r.kernel_ms); c.append(r.kernel_clock_mhz); w.append(r.workgroup_lifetime_ms)
        wall = (_t.perf_counter() - t0) / steps * 1e3
        clk = float(np.median([x for x in c if x > 0])) if any(x > 0 for x in c) else None
        return r, wall, float(np.mean(k)), clk, float(np.median(w))

    def gate(res, x_true):
        err = np.abs(res.pose - x_true); err[:, 2] = np.abs((err[:, 2] + np.pi) % (2 * np.pi) - np.pi)
        return bool(np.all(res.status == 0) and err[:, :2].max() < 1e-4 and err[:, 2].max() < 1e-4), float(err[:, :2].max()), float(err[:, 2].max())

    ctx.set_option("kernel_timing", 1)
    n_scan_mean = float(np.diff(wl.scan_offsets).mean())
    proj = api.PointNormal2fProjectorPolar(args.beams, -np.pi, np.pi, 0.3, 30.0)
    # ---- configs[4]: the same 1000 scans against a 1M-point map of the same world
    map1m = api.CloudSet(ctx, synth.make_map(world_geom, 1000000, seed=args.seed))
    al = api.MultiAligner2D(ctx, max_iterations=args.iterations, min_num_inliers=10)
    al.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8), min_num_correspondences=10))
    res, wall, k_ms, clk, wg = timed(al.prepare_batch([scan_set], [map1m], wl.x0), 2, 3, x0_sets)
    ok, em, er = gate(res, wl.x_true)
    entries.append({"config": "configs[4]: %d scans x %d-beam vs one 1000000-pt map, %d GN iters, role A, projective finder" % (args.scans, args.beams, args.iterations),
                    "value": args.scans / (wall * 1e-3), "unit": "alignments/s", "steps": 3, "warmup": 2, "ms_per_step": wall, "parity_ok": ok, "max_pose_err_m": em, "max_pose_err_rad": er,
                    "roofline": build_roofline("A", "projective", args.scans, 1000000, args.iterations, args.beams, 0.0, args.scans, n_scan_mean, k_ms, 3, clk, wg)})
    map1m.close()
    # ---- configs[3]: 65 536 loop-closure candidates = 2 048 distinct scans x 32 perturbed guesses each, Cauchy 0.05 (MULTI.json:957-962), against the 100k-point submap
    n_unique, n_cand = 2048, 65536
    wl3 = synth.make_workload(n_unique, args.map_points, seed=args.seed, n_beams=args.beams, world=world_geom, map_points=np.zeros((0, 4), np.float32))
    idx = (np.arange(n_cand) % n_unique).astype(np.int32)
    delta = synth.Stream(args.seed + 1000, salt=9).uniform(3 * n_cand, -0.05, 0.05).reshape(n_cand, 3)
    x_true3 = wl3.x_true[idx]
    x03 = synth.invert_poses(synth.compose_poses(synth.invert_poses(wl3.x_true)[idx], delta)).astype(np.float32)
    scans3 = api.CloudSet(ctx, wl3.scan_points, wl3.scan_offsets)
    map100k = api.CloudSet(ctx, synth.make_map(world_geom, args.map_points, seed=args.seed))
    al3 = api.MultiAligner2D(ctx, max_iterations=args.iterations, min_num_inliers=10)
    al3.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8), min_num_correspondences=10,
                                                                       robustifier=api.RobustifierCauchy(0.05)))
    delta_b = synth.Stream(args.seed + 2000, salt=9).uniform(3 * n_cand, -0.05, 0.05).reshape(n_cand, 3)
    x03b = synth.invert_poses(synth.compose_poses(synth.invert_poses(wl3.x_true)[idx], delta_b)).astype(np.float32)
    res, wall, k_ms, clk, wg = timed(al3.prepare_batch([scans3], [map100k], x03, fixed_index=idx[None, :]), 1, 1, [x03b, x03])
    ok, em, er = gate(res, x_true3)
    entries.append({"config": "configs[3] (one GPU's view): %d candidates over %d scans x %d-beam vs one %d-pt submap, Cauchy 0.05, %d GN iters" % (n_cand, n_unique, args.beams, args.map_points, args.iterations),
                    "value": n_cand / (wall * 1e-3), "unit": "alignments/s", "steps": 1, "warmup": 1, "ms_per_step": wall, "parity_ok": ok, "max_pose_err_m": em, "max_pose_err_rad": er,
                    "roofline": build_roofline("A", "projective", n_cand, args.map_points, args.iterations, args.beams, 0.05, n_unique, float(np.diff(wl3.scan_offsets).mean()), k_ms, 1, clk, wg)})
    scans3.close(); map100k.close()
    return entries


def run_stream(ctx, api, synth, torch, world_geom, map_set, map_dev, aligner, args, side, emit=True):
    """bench.py --stream: ranges in -> poses out, new scans every step, one step in flight.  See the option's help; DESIGN.md section 5 'fresh data'."""
    import gc
    ctx.set_option("kernel_timing", 0)                        # (switched on for a few synchronous launches at the end: see there)
    a0, a1 = -0.75 * np.pi, 0.75 * np.pi                      # the 270-degree scanner of synth.make_scans
    nb, n, nbatch = args.beams, args.scans, max(2, args.stream_batches)
    pre = api.RawDataPreprocessorProjective2D(ctx, range_min=0.3, range_max=30.0, voxelize_resolution=0.02, normal_point_distance=0.3, normal_min_points=5)   # MULTI.json:496-521, 845-853
    data = []
    for k in range(nbatch):
        poses = synth.sample_poses(world_geom, n, seed=args.seed + 7919 * (k + 1))
        rg = synth.make_scan_ranges(world_geom, poses, n_beams=nb, angle_min=a0, angle_max=a1, seed=args.seed + k)
        x_true, x0 = synth.initial_guesses(poses, seed=args.seed + k)
        data.append({"ranges": torch.from_numpy(np.ascontiguousarray(rg, np.float32)).pin_memory(), "x0": x0.astype(np.float32), "x_true": x_true})
    # the synchronous path on every distinct batch: what each streamed step must reproduce BIT FOR BIT
    for d in data:
        pre.setRawData(d["ranges"], a0, a1, 0.0, 30.0)
        fx = pre.compute()
        d["want"] = aligner.compute_batch([fx], [map_set], d["x0"])
        d["points"] = int(fx.n_points)
        fx.close()
    ahead = 1 if args.stream_ahead else 0      # 1: the scans of step i + 1 are refilled (a third scan set) behind begin(i): their preprocessing has a whole launch to hide under
    nsets = 2 + ahead
    sets = []
    for k in range(nsets):
        pre.setRawData(data[k % nbatch]["ranges"], a0, a1, 0.0, 30.0); sets.append(pre.compute())
    prep = [aligner.prepare_batch([sets[k]], [map_set], data[k % nbatch]["x0"]) for k in range(nsets)]
    state = {"i": 0, "bad": 0, "checked": 0, "last": None}

    def check(step_i, res):
        d = data[step_i % nbatch]
        same = np.array_equal(res.pose, d["want"].pose) and np.array_equal(res.information, d["want"].information) and np.array_equal(res.status, d["want"].status)
        state["checked"] += 1; state["bad"] += 0 if same else 1

    def step(timed=False):
        i = state["i"]; d = data[i % nbatch]
        if not ahead or i == 0:
            pre.setRawData(d["ranges"], a0, a1, 0.0, 30.0)
            pre.refill(sets[i % nsets])
        prep[i % nsets].set_init_poses(d["x0"])
        prep[i % nsets].begin()
        if ahead:                    # (the set batch i - 2 read: waited for in the previous step)
            pre.setRawData(data[(i + 1) % nbatch]["ranges"], a0, a1, 0.0, 30.0)
            pre.refill(sets[(i + 1) % nsets])
        res = None
        if i > 0:
            res = prep[(i - 1) % nsets].wait()
            check(i - 1, res)
        state["i"] = i + 1
        return res

    gc.collect(); gc.freeze()
    spin = 0; t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_s:
        step(); spin += 1
    for _ in range(args.warmup):
        step()
    gc.disable()
    kernel_ms, clock_mhz, wg_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):      # every pass begins one batch and retires one: K batches per K steps, one in flight across the region's edges
        step()
    elapsed = time.perf_counter() - t0
    last_i = state["i"] - 1
    res = prep[last_i % nsets].wait(); check(last_i, res)
    gc.enable()
    # the resident-input step of the default line, same process, same clock state: the ratio the verdict asks for
    d0 = data[0]
    pre.setRawData(d0["ranges"], a0, a1, 0.0, 30.0); fx = pre.compute()
    resident = aligner.prepare_batch([fx], [map_set], d0["x0"])
    for _ in range(20):
        resident.run()
    t1 = time.perf_counter()
    n_res = max(20, args.steps)
    for _ in range(n_res):
        r0 = resident.run()
    resident_ms = (time.perf_counter() - t1) / n_res * 1e3
    # the launch ALONE (HIP events, in-kernel clock): two streamed launches overlap on the chip and have no duration of their own (1.0-1.2 ms each, two at a time),
    # so the streamed loop runs untimed and the roofline block below describes k_align on these scans one launch at a time
    ctx.set_option("kernel_timing", 1)
    for _ in range(40):
        r0 = resident.run()
        kernel_ms.append(r0.kernel_ms); clock_mhz.append(r0.kernel_clock_mhz); wg_ms.append(r0.workgroup_lifetime_ms)
    ctx.set_option("kernel_timing", 0)
    est_skipped = not bool(ctx.get_option("last_cull_estimate"))
    # gates: every streamed step bit-identical to the synchronous path; poses near the generating ones (PCA normals on 2 cm voxels: centimetres, not 1e-4: DESIGN 5)
    w = data[last_i % nbatch]["want"]; xt = data[last_i % nbatch]["x_true"]
    ok_mask = w.status == 0
    err = np.abs(w.pose - xt); err[:, 2] = np.abs((err[:, 2] + np.pi) % (2 * np.pi) - np.pi)
    p99_m = float(np.percentile(err[ok_mask][:, :2].max(1), 99)); p99_rad = float(np.percentile(err[ok_mask][:, 2], 99))
    near = bool(ok_mask.mean() > 0.98 and p99_m < 3e-2 and p99_rad < 1e-2)
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else None
    clk = float(np.median([c for c in clock_mhz if c > 0])) if any(c > 0 for c in clock_mhz) else None
    ms = elapsed / args.steps * 1e3
    rbytes = 4.0 * nb * n
    out = {"metric": "scan-to-map alignments/sec (1081-beam vs 100k-pt map, 20 GN iters)", "value": n * args.steps / elapsed, "unit": "alignments/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "spinup_steps": spin, "ms_per_step": ms, "timed_region_s": elapsed, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "STREAM: every step %d NEW %d-beam range vectors (pinned host memory) -> preprocessed on the device (2 cm voxels, sliding-window normals) -> aligned vs one "
                                  "%d-pt map, %d GN iters, role A, projective finder; one step in flight, %s" % (
                                      n, nb, args.map_points, args.iterations, "scans refilled a step ahead (three sets)" if ahead else "two scan sets"),
                      "alignments_per_gpu": n, "map_points": args.map_points, "beams": nb, "iterations": args.iterations, "distinct_batches": nbatch, "refill_ahead": ahead,
                      "points_per_batch_after_preprocessing": data[0]["points"]},
           "parity_ok": bool(state["bad"] == 0 and near), "steps_checked_bitwise_against_the_synchronous_calls": state["checked"], "steps_that_differed": state["bad"],
           "max_pose_err_m": float(err[ok_mask][:, :2].max()), "max_pose_err_rad": float(err[ok_mask][:, 2].max()), "p99_pose_err_m": p99_m, "p99_pose_err_rad": p99_rad,
           "alignments_succeeded_frac": float(ok_mask.mean()),
           "parity_gate": "every streamed step BITWISE equal to lsm2d_preprocess_scans + lsm2d_align_batch on the same ranges; 99 % of the poses within 3e-2 m / 1e-2 rad of the generating "
                          "ones (the clouds are PCA normals on 2 cm voxels, MULTI.json:496-521: centimetres at corners, not the 1e-4 of analytic normals)",
           "stream": {"h2d_bytes_per_step": rbytes, "h2d_GBs_sustained": rbytes / (ms * 1e-3) / 1e9, "resident_input_ms_per_step_same_scans": resident_ms,
                      "resident_step_skips_the_estimate": est_skipped, "sustained_over_resident": resident_ms / ms,
                      "note": "resident = the same preprocessed scans already in HBM, lsm2d_align_batch per step (what the default line times); sustained_over_resident = its ms per step / "
                              "the streamed ms per step"},
           # (no committed counters for this workload -- the preprocessed scans are other clouds than the resident line's: the PMC-derived fields stay null, without a warning)
           "roofline": build_roofline("A", "projective", n, args.map_points, args.iterations, nb, 0.0, n, data[0]["points"] / float(n), k_ms or float("nan"), len(kernel_ms), clk,
                                      float(np.median(wg_ms)) if wg_ms else None, use_counters=False)}
    out["roofline"]["kernel_ms_is"] = "k_align on these scans ONE LAUNCH AT A TIME (40 synchronous launches after the streamed region): streamed launches overlap two at a time"
    out["roofline"]["counters_note"] = ("no PMC pass was taken on these clouds (preprocessed scans: PCA normals on 2 cm voxels, another survivor set than the resident line's analytic "
                                        "scans): achieved / frac stay null rather than being priced with the resident line's instruction counts; launch time and clock are measured")
    for s_ in sets:
        s_.close()
    fx.close()
    if not emit:
        return out
    print(json.dumps(out), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250, help="timed steps (default: >= 0.5 s of timed region at ~2 ms per step)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--event-every", type=int, default=1, help="HIP events + in-kernel clock stamps around the launch of every N-th timed step (1 = every step)")
    ap.add_argument("--spinup-s", type=float, default=0.3, help="seconds of untimed steps before the warm-up steps (clock ramp); 0 = none")
    ap.add_argument("--scans", type=int, default=1000, help="alignments per GPU per step")
    ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20)
    ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--cpu-sample", type=int, default=1000, help="alignments timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stream", action="store_true",
                    help="fresh data every step (VERDICT r4 item 3): each step uploads a NEW batch of raw range vectors (pinned host memory, 4 bytes per beam), preprocesses "
                         "them on the device into one of three scan sets in rotation (lsm2d_preprocess_scans_refill) and aligns them (lsm2d_align_batch_begin / _wait) "
                         "while the previous step's batch is still in flight; value = alignments/s sustained from ranges to poses")
    ap.add_argument("--stream-ahead", type=int, default=1, help="--stream: 1 (default) = three scan sets, the NEXT step's scans are refilled right behind this step's begin (their "
                                                                "preprocessing has a whole launch to hide under: include/lsm2d.h at lsm2d_align_batch_begin); 0 = two sets, refill just before begin")
    ap.add_argument("--stream-batches", type=int, default=4, help="--stream: distinct range batches cycled through (each has its own truth; every step is gated)")
    ap.add_argument("--no-streamed", action="store_true", help="skip the `streamed` block (the --stream pipeline in short, after the timed region)")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the `pipelined` block (the same resident step with two batches in flight, after the timed region)")
    ap.add_argument("--no-also", action="store_true", help="the default N=1 line carries an `also` block -- BASELINE configs[4] (1000 scans vs a 1M-point map, 3 steps) and "
                                                              "configs[3] (65 536 candidates over 2 048 scans, Cauchy 0.05, 1 step), each with its own parity gate, kernel ms, clock and roofline; this skips it")
    ap.add_argument("--pose-sets", type=int, default=8,
                    help="distinct sets of start poses the timed steps cycle through (same scans, new initial guesses of the same spread): every step uploads its start poses and "
                         "makes its placement afresh, as a tracker or a candidate sweep does.  1 = the same poses every step (the library then keeps the placement and the input "
                         "block of the unchanged batch: rounds 4-5's headline, now the `same_poses_every_step` block)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--role", choices=["A", "B"], default="A", help="A: fixed=scan, moving=map (reference tracker wiring); B: fixed=map, moving=scan")
    ap.add_argument("--finder", choices=["projective", "nn", "kdtree", "distmap"], default="projective",
                    help="projective: CorrespondenceFinderProjective2f (reference default); nn: CorrespondenceFinderKDTree2D with the exact grid search; "
                         "kdtree: CorrespondenceFinderKDTree2D with the reference's own tree and single-leaf descent (max_leaf_range 1e-2, min_leaf_points 20); "
                         "distmap: CorrespondenceFinderNN2D (row f4)")
    ap.add_argument("--resolution", type=float, default=0.05, help="distance-map finder: metres per pixel")
    ap.add_argument("--max-distance", type=float, default=0.5, help="NN finder gate [m]")
    ap.add_argument("--cauchy", type=float, default=0.0, help="Cauchy chi_threshold (0 = no robustifier); configs[3] uses 0.05 (MULTI.json:957-962)")
    ap.add_argument("--total-candidates", type=int, default=0,
                    help="BASELINE configs[3]: a fixed sweep of this many candidate alignments sharded over the ranks (strong scaling); "
                         "overrides --scans with this rank's share and gathers the poses on every rank at the end")
    ap.add_argument("--unique-scans", type=int, default=0, help="ray-cast only this many scans; candidates reuse them through an index array (loop-closure sweep)")
    ap.add_argument("--shard-by", choices=["work", "count"], default="work",
                    help="--total-candidates: shard the sweep over the ranks by estimated work (lsm2d_estimate_work: chunks of the map each candidate's first "
                         "iteration streams) or by candidate count")
    args = ap.parse_args()

    # `python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): this process has not touched the GPU yet -- it becomes the
    # launcher itself: N fresh rank processes through torch.distributed.run, exactly as the driver starts them, rank 0's JSON line relayed, its exit
    # code ours.  (Until round 3 this case silently ran ONE rank and printed n_gpus: 1.)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        import torch                       # (importing torch / counting devices does not initialise the GPU)
        n_dev = torch.cuda.device_count()
        if os.environ.get("LSM2D_BENCH_BACKEND", "nccl") == "nccl" and n_dev < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d device(s) visible (RCCL wants one device per rank; LSM2D_BENCH_BACKEND=gloo lets ranks share a card)" % (args.gpus, n_dev))
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        child = subprocess.run(cmd)        # stdout / stderr inherited: rank 0's line is the only JSON line
        raise SystemExit(child.returncode)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line's n_gpus must be the number of ranks that ran")
    # N > 1: pin this rank's host thread to its own slice of the allowed cores BEFORE anything touches the GPU (the runtime's helper threads inherit the mask),
    # and start its note file
    affinity = sorted(os.sched_getaffinity(0))
    if world > 1:
        mine = rank_affinity(affinity, local_rank, world)
        try:
            os.sched_setaffinity(0, mine); affinity = mine
        except OSError:
            pass
        note_dir = os.environ.get("LSM2D_BENCH_RANK_DIR") or (os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        if note_dir:
            _RANK_NOTE["path"] = os.path.join(note_dir, "bench_rank%d.json" % rank)
        rank_note("init", rank=rank, local_rank=local_rank, world=world, pid=os.getpid(), cpu_affinity=affinity, argv=sys.argv[1:])
    # rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (tests): LSM2D_BENCH_BACKEND=gloo lets several ranks share a card
    # (RCCL refuses two ranks on one device); the ranks then run the same kernels, shards and cross-rank check, only the transport differs
    backend = os.environ.get("LSM2D_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or bool(os.environ.get("LSM2D_BENCH_FORCE_DIST"))   # the env var rehearses the RCCL path on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ranks_seen = dist.get_world_size() if use_dist else 1

    from srrg2_laser_slam_2d_amd import api, distributed, synth

    strong = args.total_candidates > 0
    if strong:
        lo, hi = distributed.shard_range(args.total_candidates, rank, world)      # (refined below by estimated work: --shard-by)
        args.scans = hi - lo

    # ---- inputs: the shared local map comes from rank 0 (RCCL broadcast), each rank ray-casts its own scans
    world_geom = synth.make_world(args.seed)
    map_dev = distributed.broadcast_map(
        synth.make_map(world_geom, args.map_points, seed=args.seed) if rank == 0 else None, args.map_points, local_rank)
    # strong scaling: ONE global candidate list (every rank derives the same one from the seed) cut into this rank's shard; weak: every rank its own scans
    n_list = args.total_candidates if strong else args.scans
    n_unique = args.unique_scans if 0 < args.unique_scans < n_list else n_list
    wl = synth.make_workload(n_unique, args.map_points, seed=args.seed, n_beams=args.beams, pose_seed_offset=0 if strong else rank,
                             world=world_geom, map_points=np.zeros((0, 4), np.float32))
    scan_index = None
    if n_unique < n_list:      # candidate i = (scan i mod n_unique, its own perturbed initial guess)
        scan_index = (np.arange(n_list) % n_unique).astype(np.int32)
        st = synth.Stream(args.seed + 1000 + (0 if strong else rank), salt=9)
        delta = st.uniform(3 * n_list, -0.05, 0.05).reshape(n_list, 3)
        t_true = synth.invert_poses(wl.x_true)[scan_index]
        wl.x_true = wl.x_true[scan_index]
        wl.x0 = synth.invert_poses(synth.compose_poses(t_true, delta)).astype(np.float32)
    elif strong:
        scan_index = np.arange(n_list, dtype=np.int32)
    # a dedicated torch stream, made current: the kernels, the HIP events around them and torch's own view all sit on it
    # (the legacy default stream synchronises device-wide, which doubles the per-call latency of the single-scan config)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=local_rank)
    torch.cuda.set_stream(side)
    ctx = api.Context(local_rank, stream=side.cuda_stream)
    options_set = {}
    for kv in filter(None, os.environ.get("LSM2D_BENCH_OPTIONS", "").split(",")):      # tuning experiments: context options as key=value pairs (printed in the line)
        k_opt, _, v_opt = kv.partition("=")
        ctx.set_option(k_opt.strip(), int(v_opt)); options_set[k_opt.strip()] = int(v_opt)
    map_set = api.CloudSet(ctx, map_dev)                      # stays in HBM, no host copy
    scan_set = api.CloudSet(ctx, wl.scan_points, wl.scan_offsets)
    if args.finder == "projective":
        proj = api.PointNormal2fProjectorPolar(args.beams, -np.pi, np.pi, 0.3, 30.0)
        finder = api.CorrespondenceFinderProjective2f(ctx, proj, point_distance=0.5, normal_cos=0.8)
    elif args.finder in ("nn", "kdtree"):
        finder = api.CorrespondenceFinderKDTree2D(ctx, max_distance_m=args.max_distance, normal_cos=0.8, search="exact" if args.finder == "nn" else "kdtree")
    else:
        finder = api.CorrespondenceFinderNN2D(ctx, max_distance_m=args.max_distance, resolution=args.resolution, normal_cos=0.8)
    aligner = api.MultiAligner2D(ctx, max_iterations=args.iterations, min_num_inliers=10)
    robust = api.RobustifierCauchy(args.cauchy) if args.cauchy > 0 else None
    aligner.param_slice_processors.append(api.AlignerSliceProcessorLaser2D(finder, min_num_correspondences=10, robustifier=robust))
    if args.stream:
        if world != 1 or args.role != "A" or args.finder != "projective" or strong:
            raise SystemExit("--stream: one GPU, role A, projective finder")
        run_stream(ctx, api, synth, torch, world_geom, map_set, map_dev, aligner, args, side)
        ctx.close()
        return
    shard_info = None
    if strong:
        # this rank's shard of the global list: by estimated work (lsm2d_estimate_work -- with the exact culling an alignment's time follows the number
        # of map chunks it streams) or by count.  Every rank computes the same estimates, hence the same cuts: no collective.
        shards = [distributed.shard_range(n_list, r, world) for r in range(world)]
        if args.shard_by == "work" and args.role == "A" and args.finder == "projective":
            work = aligner.estimate_work([scan_set], [map_set], wl.x0, fixed_index=scan_index[None, :])
            shards = distributed.shard_by_work(work, world)
            tot = [float(work[a:b].sum()) for a, b in shards]
            shard_info = {"by": "work", "candidates_per_rank": [b - a for a, b in shards], "work_max_over_mean": max(tot) / (sum(tot) / world) if sum(tot) > 0 else 1.0,
                          "work_max_over_mean_if_sharded_by_count": (lambda t: max(t) / (sum(t) / world))([float(work[a:b].sum()) for a, b in
                                                                                                            [distributed.shard_range(n_list, r, world) for r in range(world)]])}
        else:
            shard_info = {"by": "count", "candidates_per_rank": [b - a for a, b in shards]}
        lo, hi = shards[rank]
        args.scans = hi - lo
        wl.x0 = wl.x0[lo:hi]; wl.x_true = wl.x_true[lo:hi]; scan_index = scan_index[lo:hi]
    # Round 6: the timed steps cycle through --pose-sets distinct sets of START POSES for the same scans (set 0: the workload's own; set k: new initial guesses
    # T0 = T* . v2t(delta), delta ~ U(-0.05, 0.05)^3 from another stream) -- a step whose start poses are those of the step before is a step no consumer submits, and
    # the library answers it from what it kept (no upload, no placement estimate).  Every set is gated before the clock starts.
    n_sets = max(1, args.pose_sets)
    t_true = synth.invert_poses(wl.x_true)
    x0_sets_a = [wl.x0.astype(np.float32)]
    for k in range(1, n_sets):
        dk = synth.Stream(args.seed + 104729 * k + (0 if strong else rank), salt=6).uniform(3 * len(t_true), -0.05, 0.05).reshape(len(t_true), 3)
        x0_sets_a.append(synth.invert_poses(synth.compose_poses(t_true, dk)).astype(np.float32))
    if args.role == "A":
        x0_sets, x_true = x0_sets_a, wl.x_true
        idx = None if scan_index is None else scan_index[None, :]

        def make_prepared(x_start=None):
            return aligner.prepare_batch([scan_set], [map_set], x0_sets[0] if x_start is None else x_start, fixed_index=idx)      # descriptor and result arrays built once: the step is the C-ABI call
    else:                           # the estimate is scan-in-map
        x0_sets = [synth.invert_poses(x.astype(np.float64)).astype(np.float32) for x in x0_sets_a]; x_true = synth.invert_poses(wl.x_true)
        idx = None if scan_index is None else scan_index[None, :]

        def make_prepared(x_start=None):
            return aligner.prepare_batch([map_set], [scan_set], x0_sets[0] if x_start is None else x_start, moving_index=idx)
    x0 = x0_sets[0]
    prepared = make_prepared()
    step_state = {"i": 0}

    def step():
        k = step_state["i"] % n_sets; step_state["i"] += 1
        if n_sets > 1:
            prepared.set_init_poses(x0_sets[k])      # 12 bytes per alignment written in place; the call below uploads them
        return prepared.run()

    tol_m, tol_rad = (1e-4, 1e-4) if args.finder == "projective" else (5e-3, 2e-3)      # (the point-query finders match discrete map points: millimetres from the generating pose -- their 1e-4 gate is against the oracle, below)

    def gate(r):
        e = np.abs(r.pose - x_true); e[:, 2] = np.abs((e[:, 2] + np.pi) % (2 * np.pi) - np.pi)
        return bool(np.all(r.status == 0) and e[:, :2].max() < tol_m and e[:, 2].max() < tol_rad), e
    sets_ok = True; want_sets = []
    for k in range(n_sets):          # every pose set once, synchronously: its gate, and the results the pipelined block compares with
        prepared.set_init_poses(x0_sets[k]); rk = prepared.run(copy=True)
        sets_ok = sets_ok and gate(rk)[0]; want_sets.append(rk)
    prepared.set_init_poses(x0_sets[0])

    # the interpreter's cyclic collector walks every object torch has imported (tens of milliseconds, once or twice per few hundred
    # steps: one such pause was 8 % of a 0.46 s timed region).  Collect now and park what exists in the permanent generation, so
    # the collector only ever looks at what the loop itself allocates -- nothing is switched off.  BEFORE the warm-up, not behind it: the
    # chip drops its clock within an idle stretch of that length and needs ~25 launches (40 ms) of load to get it back
    # (tools/clock_trace.py, profiles/r02/clock_trace_r02k.txt: 2.06 GHz for the first 25 steps after a pause, 2.19 GHz from then on) --
    # with the collection between warm-up and timing, 20 timed steps ran entirely on the low clock.
    import gc
    rank_note("inputs", alignments=int(args.scans))
    gc.collect(); gc.freeze()
    # N > 1: the ranks line up BEFORE the clock ramp, and the collectives the timed region is bracketed by run once here.  Measured on one
    # rank with the RCCL path forced (LSM2D_BENCH_FORCE_DIST=1, --steps 20): without this the barrier in front of the timed region was the
    # first of its kind -- it loaded RCCL's kernels while the GPU sat idle, the chip dropped its clock and the 20 timed steps ran 8 % slow
    # (1.092 vs 0.996 ms per step, kernel 0.985 vs 0.913 ms); and ranks that start their ramp at different times reach that barrier at
    # different times, the early ones idling there just as long.
    if use_dist:
        step()
        torch.cuda.synchronize()
        dist.barrier()
        warm = torch.zeros(1, dtype=torch.float64, device="cuda")
        dist.all_reduce(warm, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
    # clock ramp, untimed: the same step for --spinup-s seconds (the count is printed as `spinup_steps`), then the W warm-up steps
    spinup_steps = 0
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup_s:
        step(); spinup_steps += 1
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    rank_note("warmup", spinup_steps=spinup_steps)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    rank_note("timed")
    gc_was_on = gc.isenabled()
    if os.environ.get("LSM2D_BENCH_GC_OFF", "1") != "0":
        gc.disable()                 # as timeit does: the collector's pauses are the interpreter's, not the measured path's (re-enabled below)
    t0 = time.perf_counter()
    kernel_ms = []; clock_mhz = []; wg_ms = []; step_s = []
    # The kernel's own duration is taken live, inside the timed region, on every --event-every-th step (default: every step; measured: timing
    # every step or every fourth makes no difference to a 1000-alignment step, 1.539 vs 1.534 ms).  Steps 0, e, 2e, ... are timed.
    # strong scaling (a fixed sweep sharded over the ranks): the sweep's consumer wants every candidate's pose, so the all_gather of 12 bytes per
    # candidate is PART OF THE STEP and is timed with it (round 2 gathered once, behind the timed region)
    gather_pad = 0
    if strong and use_dist:
        gather_pad = max(h - l for l, h in shards)
        gather_buf = np.zeros((gather_pad, 3), np.float32)
    gathered = None
    for i_step in range(args.steps):
        timed_step = i_step % args.event_every == 0
        if ctx.kernel_timing != timed_step:
            ctx.set_option("kernel_timing", int(timed_step))
        ts = time.perf_counter()
        res = step()
        if gather_pad:
            gather_buf[: len(res.pose)] = res.pose
            gathered = distributed.gather_results(gather_buf)
        step_s.append(time.perf_counter() - ts)
        if timed_step:
            kernel_ms.append(res.kernel_ms)          # HIP events around the k_align launch, on the launch stream
            clock_mhz.append(res.kernel_clock_mhz)   # s_memtime / s_memrealtime stamps inside the same launch
            wg_ms.append(res.workgroup_lifetime_ms)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    rank_note("timed_done", ms_per_step=elapsed / args.steps * 1e3, steps=args.steps)
    if gc_was_on:
        gc.enable()
    if gather_pad:
        assert gathered is not None and gathered.shape == (gather_pad * world, 3)
    per_rank_ms = None
    if use_dist:
        mine_ms = elapsed / args.steps * 1e3
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's own ms per step (barrier to barrier), rank order: the spread shows which rank the maximum waits for
        pr = distributed.gather_results(np.array([[mine_ms]], np.float32))
        per_rank_ms = [float(v) for v in np.asarray(pr).ravel()]

    # correctness gate: a timing only counts if the poses are right (noise-free data -> generating pose): the last timed step's, and every pose set's (above)
    ok, err = gate(res)
    ok = ok and sets_ok
    last_set = (step_state["i"] - 1) % n_sets
    if use_dist:
        f = torch.tensor([1 if ok else 0], device="cuda"); dist.all_reduce(f, op=dist.ReduceOp.MIN); ok = bool(f.item())

    # cross-rank check: every rank's first 16 candidates (scans, initial guesses) and the poses it got for them are gathered, and rank 0
    # aligns each rank's 16 alone on its own GPU: the poses must agree BIT FOR BIT -- sharding changes where an alignment runs, never
    # its result
    cross = None
    if use_dist:
        def align_alone(clouds, x0_r):
            offs_r = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int32)
            set_r = api.CloudSet(ctx, np.concatenate(clouds, 0), offs_r)
            rr = (aligner.compute_batch([set_r], [map_set], x0_r) if args.role == "A" else aligner.compute_batch([map_set], [set_r], x0_r))
            set_r.close()
            return rr.pose
        chk = distributed.cross_rank_check(wl.scan_points, wl.scan_offsets, scan_index, x0_sets[last_set], res.pose, args.beams, align_alone)
        if rank == 0:
            same, nw, ncheck = chk
            cross = "%d of %d ranks: first %d poses bit-identical to rank 0 aligning the same candidates alone" % (same, nw, ncheck)
            ok = ok and same == nw

    if rank == 0:
        n_total = (args.total_candidates if strong else args.scans * world) * args.steps
        n_scan_mean = float(np.diff(wl.scan_offsets).mean())
        bytes_per_alignment = algorithmic_bytes_per_alignment(args.role, args.finder, args.map_points, n_scan_mean, args.beams, args.iterations)
        k_ms = float(np.mean(kernel_ms))
        clk = float(np.median([c for c in clock_mhz if c > 0])) if any(c > 0 for c in clock_mhz) else None
        default_cfg = (args.role, args.finder, args.scans, args.map_points, args.iterations, args.beams, args.cauchy, n_unique) == \
                      ("A", "projective", 1000, 100000, 20, 1081, 0.0, 1000)
        cfg_key = "role%s/%s/scans%d/map%d/it%d/beams%d" % (args.role, args.finder, args.scans, args.map_points, args.iterations, args.beams)
        if args.cauchy > 0:                       # a robustified run is another instruction stream (the log, the weights): its own counters
            cfg_key += "/cauchy%g" % args.cauchy
        if n_unique != args.scans:                # scans shared through the index array: another memory pattern
            cfg_key += "/unique%d" % n_unique
        roof = build_roofline(args.role, args.finder, args.scans, args.map_points, args.iterations, args.beams, args.cauchy, n_unique, n_scan_mean,
                              k_ms, len(kernel_ms), clk, float(np.median(wg_ms)) if wg_ms else None)
        out = {
            "metric": "scan-to-map alignments/sec (1081-beam vs 100k-pt map, 20 GN iters)",
            "value": n_total / elapsed, "unit": "alignments/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "spinup_steps": spinup_steps,
            "ms_per_step": elapsed / args.steps * 1e3, "timed_region_s": elapsed,
            "ms_per_step_median_rank0": float(np.median(step_s)) * 1e3, "ms_per_step_max_rank0": float(np.max(step_s)) * 1e3, "slowest_step_rank0": int(np.argmax(step_s)), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s%d scans/GPU x %d-beam vs one %d-pt map, %d GN iters, role %s (%s), %s finder%s"
                                   % ("configs[1]: " if default_cfg else "", args.scans, args.beams, args.map_points, args.iterations, args.role,
                                      "fixed=scan, moving=map" if args.role == "A" else "fixed=map, moving=scan", args.finder,
                                      ", Cauchy tau %g" % args.cauchy if args.cauchy > 0 else ""),
                       "unique_scans": n_unique, "pose_sets": n_sets,
                       "alignments_per_gpu": args.scans, "map_points": args.map_points, "beams": args.beams,
                       "iterations": args.iterations, "parallelism": "alignments sharded, map replicated (RCCL broadcast)"},
            "parity_ok": ok, "max_pose_err_m": float(err[:, :2].max()), "max_pose_err_rad": float(err[:, 2].max()),
            "roofline": roof,
        }
        if options_set:
            out["options"] = options_set
        # what the timed steps did about the placement of the batch: the library keeps the order it made for a batch and launches no estimate (k_cull_estimate, ~33 us)
        # when the SAME batch -- sets, indices, parameters, start poses -- is run again, which is what this resident-input step does; `--stream` (fresh scans
        # every step) pays it every time
        out["placement"] = {"estimate_launched_in_last_step": bool(ctx.get_option("last_cull_estimate")), "pose_sets": n_sets,
                            "note": "the timed steps cycle through %d sets of start poses: every step uploads its poses and makes its placement afresh (a batch run again with "
                                    "unchanged sets and start poses keeps both: the same_poses_every_step block)" % n_sets}

        def same_poses_block():
            # rounds 4-5's headline: the SAME start poses every step -- the library keeps the unchanged batch's placement (no k_cull_estimate launch) and input block (no upload)
            prepared.set_init_poses(x0_sets[0])
            for _ in range(max(args.warmup, 3)):
                prepared.run()
            ctx.set_option("kernel_timing", 1)
            t1 = time.perf_counter(); km = []
            for _ in range(args.steps):
                km.append(prepared.run().kernel_ms)
            dt = time.perf_counter() - t1
            return {"value": args.scans * args.steps / dt, "unit": "alignments/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "kernel_ms": float(np.mean(km)),
                    "estimate_launched_in_last_step": bool(ctx.get_option("last_cull_estimate")),
                    "note": "what no consumer submits (identical start poses step after step); kept beside the headline because rounds 4-5 reported it as the headline"}
        # The extras of the default line (profiler passes run --no-also).  None of them may cost the headline its line: a failure inside one is reported IN its block.
        def extra(name, fn):
            try:
                out[name] = fn()
            except Exception as e:      # noqa: BLE001 -- whatever it is, the line above it has been measured and must still be printed
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
                try:
                    ctx.synchronize()
                except Exception:
                    pass
        if world == 1 and n_sets > 1 and not args.no_also and not options_set:
            extra("same_poses_every_step", same_poses_block)
        if world == 1 and default_cfg and not args.no_also and not options_set:
            def sum_order_block():
                # the reference's order of summation ("sum_order" 1: pair after pair, bitwise the sequential fp32 oracle -- checked in the cpu_baseline leg below): its price
                ctx.set_option("sum_order", 1)
                try:
                    ps = make_prepared()
                    keep = []
                    for k in range(n_sets):
                        ps.set_init_poses(x0_sets[k]); rk = ps.run(copy=True)
                        if k == 0:
                            keep = [rk.pose[:16].copy(), rk.information[:16].copy(), rk.status[:16].copy()]
                        if not gate(rk)[0]:
                            raise RuntimeError("sum_order 1: pose gate failed for pose set %d" % k)
                    ctx.set_option("kernel_timing", 1)
                    t1 = time.perf_counter(); km = []
                    for i_ in range(args.steps):
                        ps.set_init_poses(x0_sets[i_ % n_sets]); km.append(ps.run().kernel_ms)
                    dt = time.perf_counter() - t1
                finally:
                    ctx.set_option("sum_order", 0)
                return {"value": args.scans * args.steps / dt, "unit": "alignments/s", "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "kernel_ms": float(np.mean(km)),
                        "kernel": "k_align_seq<1,0,0,0,5>", "over_default_order": (dt / args.steps * 1e3) / (elapsed / args.steps * 1e3), "parity_ok": True, "_first16": keep,
                        "note": "lsm2d_set_option(\"sum_order\", 1): H, b and chi^2 added pair after pair in the reference's order (nicp_post.m:69-90) instead of in trees"}
            extra("sum_order_1", sum_order_block)
        if world == 1 and default_cfg and not args.no_also and not args.no_pipelined and not options_set:
            extra("pipelined", lambda: measure_pipelined(ctx, prepared, make_prepared(), x0_sets, want_sets, args, roof))
        if world == 1 and default_cfg and not args.no_also and not args.no_streamed and not options_set:
            # the streamed pipeline of `--stream` in short (fresh ranges every step, 300 steps): so that the default line -- the one the round-end driver records -- has timed it
            import copy
            a2 = copy.copy(args); a2.steps, a2.warmup, a2.spinup_s = 300, 5, 0.1
            kt = ctx.get_option("kernel_timing")

            def streamed_block():
                try:
                    so = run_stream(ctx, api, synth, torch, world_geom, map_set, map_dev, aligner, a2, side, emit=False)
                finally:
                    ctx.set_option("kernel_timing", kt)
                return {"value": so["value"], "unit": so["unit"], "ms_per_step": so["ms_per_step"], "steps": so["steps"], "workload": so["config"]["workload"],
                        "sustained_over_resident": so["stream"]["sustained_over_resident"], "resident_input_ms_per_step_same_scans": so["stream"]["resident_input_ms_per_step_same_scans"],
                        "h2d_GBs_sustained": so["stream"]["h2d_GBs_sustained"], "steps_checked_bitwise_against_the_synchronous_calls": so["steps_checked_bitwise_against_the_synchronous_calls"],
                        "steps_that_differed": so["steps_that_differed"], "parity_ok": so["parity_ok"], "parity_gate": so["parity_gate"]}
            extra("streamed", streamed_block)
        if world == 1 and default_cfg and not args.no_also and not options_set:
            extra("also", lambda: measure_also(ctx, api, synth, world_geom, wl, scan_set, args, x0_sets))
        if cross:
            out["cross_rank_check"] = cross
        if world > 1:
            out["cpu_affinity_rank0"] = affinity
            out["rank_notes"] = _RANK_NOTE["path"] and os.path.join(os.path.dirname(_RANK_NOTE["path"]), "bench_rank<r>.json")
        if per_rank_ms is not None:
            out["ms_per_step_per_rank"] = per_rank_ms
            out["ms_per_step_rank_max"] = max(per_rank_ms); out["ms_per_step_rank_min"] = min(per_rank_ms)
        if shard_info is not None:
            out["sharding"] = shard_info
        if gather_pad:
            out["strong_scaling_gather"] = "all_gather of %d x 12 B per rank inside every timed step" % gather_pad
        if world == 1 and not args.no_cpu_baseline:
            from oracle import pyoracle as po       # the checker, timed as the CPU baseline ("port")
            ns = min(args.cpu_sample, n_unique)
            offs = wl.scan_offsets[: ns + 1]
            map_host = map_dev.cpu().numpy()
            osp = po.slice_params(finder={"projective": po.FINDER_PROJECTIVE, "nn": po.FINDER_NN, "distmap": po.FINDER_DISTMAP, "kdtree": po.FINDER_KDTREE_APPROX}[args.finder],
                                  canvas_cols=args.beams, max_distance=args.max_distance, resolution=args.resolution,
                                  **({"robustifier": po.ROBUST_CAUCHY, "chi_threshold": args.cauchy} if args.cauchy > 0 else {}))
            po.lib()                                   # load (or build) the checker before the clock starts
            t1 = time.perf_counter()
            if args.role == "A":
                xo, _, st, _ = po.align_batch(po.aligner_params(args.iterations), osp, wl.scan_points[: offs[-1]], offs, map_host, x0[:ns], n_threads=1)
            else:
                xo = np.empty((ns, 3), np.float32)
                for i in range(ns):
                    r = po.align(po.aligner_params(args.iterations), [osp], [map_host], [wl.scan_points[offs[i]:offs[i + 1]]], x0[i])
                    xo[i] = r["pose"]
            cpu_s = time.perf_counter() - t1
            cpu = host_cpu()
            all_cores = None
            if args.role == "A" and ns >= 64:       # BASELINE.md section 3, second row: the host's cores, the workers sharing one atomic work counter
                # as many threads as the process may RUN at once: its affinity mask capped by the cgroup's CPU quota (a 256-thread mask with a 16-CPU quota -- this pool's
                # GPU boxes -- is throttled to 16 threads' worth whatever is started: round 5's row, 256 threads, read 10 x one thread for that reason)
                quota = cpu["cgroup_cpu_quota"]
                nt = max(1, min(cpu["threads_allowed"], int(math.ceil(quota)) if quota else cpu["threads_allowed"]))
                tt = {}
                t2 = time.perf_counter()
                po.align_batch(po.aligner_params(args.iterations), osp, wl.scan_points[: offs[-1]], offs, map_host, x0[:ns], n_threads=nt, thread_times=tt)
                wall = time.perf_counter() - t2
                all_cores = {"value": ns / wall, "threads": nt, "threads_allowed_by_affinity": cpu["threads_allowed"], "cgroup_cpu_quota": quota, "physical_cores": cpu["physical_cores"],
                             "wall_s": wall, "over_one_thread": (ns / wall) / (ns / cpu_s),
                             "thread_wall_s_min_median_max": [float(np.min(tt["seconds"])), float(np.median(tt["seconds"])), float(np.max(tt["seconds"]))],
                             "alignments_per_thread_min_max": [int(np.min(tt["alignments"])), int(np.max(tt["alignments"]))],
                             "per_alignment_ms_inside_a_thread": float(1e3 * tt["seconds"].sum() / max(int(tt["alignments"].sum()), 1))}
            res0 = want_sets[0]                      # the device's results for pose set 0: the start poses the CPU runs use
            d = np.abs(res0.pose[:ns] - xo)
            # the same checker summing in the kernels' order must reproduce the device BIT FOR BIT (a handful of alignments); with the option "sum_order" 1 set for
            # the whole run (LSM2D_BENCH_OPTIONS) the device sums in the reference's order and it is the SEQUENTIAL checker that must
            seq_run = bool(options_set.get("sum_order"))
            nbit = min(16, ns); bit_equal = 0
            for i in range(nbit):
                sc = wl.scan_points[offs[i]:offs[i + 1]]
                fx, mv = ([sc], [map_host]) if args.role == "A" else ([map_host], [sc])
                rt = po.align(po.aligner_params(args.iterations, device_order=not seq_run), [osp], fx, mv, x0[i])
                bit_equal += int(np.array_equal(res0.pose[i], rt["pose"]) and np.array_equal(res0.information[i], rt["H"]))
            if isinstance(out.get("sum_order_1"), dict) and "_first16" in out["sum_order_1"]:
                # the sum_order 1 block's first alignments against the SEQUENTIAL fp32 checker: bit for bit
                kp, kh, kst = out["sum_order_1"].pop("_first16"); same = 0
                for i in range(min(16, ns)):
                    sc = wl.scan_points[offs[i]:offs[i + 1]]
                    rs = po.align(po.aligner_params(args.iterations), [osp], [sc], [map_host], x0[i])
                    same += int(np.array_equal(kp[i], rs["pose"]) and np.array_equal(kh[i], rs["H"]) and int(kst[i]) == rs["status"])
                out["sum_order_1"]["bit_identical_to_sequential_port"] = "%d of %d alignments (status, pose and information matrix)" % (same, min(16, ns))
                out["sum_order_1"]["parity_ok"] = bool(same == min(16, ns))
            if args.finder != "projective":
                # Round 6: the point-query finders' 1e-4 gate is against the ORACLE on the same inputs (the generating pose is millimetres away for them: discrete map points)
                oracle_ok = bool(d[:, :2].max() < 1e-4 and d[:, 2].max() < 1e-4)
                out["parity_ok"] = bool(out["parity_ok"] and oracle_ok)
                out["parity_gate"] = "status 0 everywhere, within %g m / %g rad of the generating pose, and the first %d alignments within 1e-4 m / 1e-4 rad of the CPU oracle on the same inputs" % (tol_m, tol_rad, ns)
            out["cpu_baseline"] = {"value": ns / cpu_s, "unit": "alignments/s", "cores": 1, "kind": "port", "cpu_model": cpu["model"],
                                   "sample": "first %d alignments of the same batch, CPU restatement of the reference algorithm (oracle/, gcc -O3 -march=native, fp32), %.1f s"
                                             % (ns, cpu_s),
                                   "max_pose_diff_gpu_vs_cpu_m": float(d[:, :2].max()), "max_pose_diff_gpu_vs_cpu_rad": float(d[:, 2].max()),
                                   ("bit_identical_to_sequential_port" if seq_run else "bit_identical_to_device_order_port"): "%d of %d alignments (pose and information matrix)" % (bit_equal, nbit)}
            if all_cores:
                out["cpu_baseline"]["all_cores"] = all_cores
        if isinstance(out.get("sum_order_1"), dict):
            out["sum_order_1"].pop("_first16", None)      # (--no-cpu-baseline: nothing to compare with)
        print(json.dumps(out), flush=True)
    rank_note("done", parity_ok=bool(ok), max_pose_err_m=float(err[:, :2].max()))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
