"""Turn the rocprofv3 --pmc passes of one bench.py workload into profiles/counters.json (read back by bench.py).

usage (GPU box, after tools/profile_round.sh <tag>):  python tools/write_counters.py gpurun_out/<tag> [--scans 1000 --map-points 100000 ...]
The file records the sha256 of the kernel sources the counters were taken on (srrg2_laser_slam_2d_amd.build.source_hash):
bench.py reports PMC-derived numbers only while that hash matches the library it runs.
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE / WRITE_SIZE are in KB, and on gfx950 FETCH_SIZE reads half
of a wide coalesced stream (MI355X_MICROARCH.md, HBM section).
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from srrg2_laser_slam_2d_amd import build as hip_build  # noqa: E402


def pmc_means(root, want):
    acc = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        per = defaultdict(dict)
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]      # ("k_align" means the k_align<...> template itself, not k_align_seq / _narrow / _pair: round 6)
            if (want + "<" in name or want + "I" in name) if want == "k_align" else want in name:
                per[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
        for d in per.values():
            for k, v in d.items():
                acc[k].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--kernel", default="k_align")
    ap.add_argument("--role", default="A"); ap.add_argument("--finder", default="projective")
    ap.add_argument("--scans", type=int, default=1000); ap.add_argument("--map-points", type=int, default=100000)
    ap.add_argument("--iterations", type=int, default=20); ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--cauchy", type=float, default=0.0); ap.add_argument("--unique-scans", type=int, default=0)
    ap.add_argument("--tag", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "counters.json"))
    a = ap.parse_args()
    m, cnt = pmc_means(a.root, a.kernel)
    if "SQ_INSTS_VALU" not in m:
        raise SystemExit("no SQ_INSTS_VALU for %s under %s" % (a.kernel, a.root))
    key = "role%s/%s/scans%d/map%d/it%d/beams%d" % (a.role, a.finder, a.scans, a.map_points, a.iterations, a.beams)
    if a.cauchy > 0:
        key += "/cauchy%g" % a.cauchy
    if a.unique_scans and a.unique_scans != a.scans:
        key += "/unique%d" % a.unique_scans
    ent = {"valu_insts_per_launch": m["SQ_INSTS_VALU"], "launches_averaged": cnt["SQ_INSTS_VALU"], "pmc_means": m}
    if a.role == "A" and a.finder == "projective":
        # ONE transcendental (v_rsq_f32) per point slot of the lane-chunked stream since round 2's bearing = asin(min / r) (padding slots
        # fail the range gate in front of it; they are counted anyway: an upper bound of 0.4 %)
        T = -(-(-(-a.map_points // 2)) // 512)
        full = 1.0 * a.scans * a.iterations * (T * 512 * 2) / 64.0      # every point slot of the lane-chunked copy, every iteration (no culling)
        ent["wave_points_full"] = full
        # with the exact culling (round 3) the point visits are data dependent: the stream's one v_rsq_f32 per point visit is COUNTED
        # (SQ_INSTS_VALU_TRANS_F32: wave-instructions), minus the handful the culling test itself and the fixed cloud's pass issue
        visits = m.get("SQ_INSTS_VALU_TRANS_F32", full)
        ent["trans_insts_per_launch"] = visits
        ent["wave_points_per_launch"] = visits
        ent["point_visits_frac"] = visits / full
        probe = os.path.join(a.root, "valu_issue_probe.txt")      # tools/valu_issue_probe.hip run in the same pass: the stream's own issue rate
        if not os.path.exists(probe):                              # (a mode's passes live one level below the headline's: tools/pmc_modes.sh)
            probe = os.path.join(os.path.dirname(os.path.abspath(a.root)), "valu_issue_probe.txt")
        if os.path.exists(probe):
            import re
            mm = re.search(r"k_align's point stream.*?:\s*([0-9.]+) cycles of a SIMD per point", open(probe).read())
            if mm:
                ent["stream_cycles_per_wave_point"] = float(mm.group(1))
    if "SQ_INSTS_VMEM_RD" in m:
        ent["vmem_rd_insts_per_launch"] = m["SQ_INSTS_VMEM_RD"]
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        ent["fetch_size_kb"] = m["FETCH_SIZE"]; ent["write_size_kb"] = m["WRITE_SIZE"]
        ent["hbm_bytes_per_launch"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0
    try:
        cj = json.load(open(a.out))
    except (OSError, ValueError):
        cj = {}
    h = hip_build.source_hash()
    if cj.get("csrc_sha256") != h:
        cj = {"csrc_sha256": h, "configs": {}}
    cj["configs"][key] = ent
    cj["source"] = ("rocprofv3 --pmc passes (each counter group in its own run, --kernel-trace only) of `python3 bench.py --steps 3 --warmup 1 "
                    "--no-cpu-baseline`, per-launch means over the profiled k_align launches; %s" % (a.tag or a.root))
    json.dump(cj, open(a.out, "w"), indent=1)
    print("wrote", a.out, key, {k: ent[k] for k in ent if k != "pmc_means"})


if __name__ == "__main__":
    main()
