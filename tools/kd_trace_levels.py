"""Per-level durations of the KD-tree build of the LARGEST cloud in a rocprofv3 kernel trace of tools/kd_build_bench.py:
    python tools/kd_trace_levels.py <dir with */*_kernel_trace.csv> ["<1>" | "<0>"]"""
import csv, glob, os, sys
fs = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)
rows = list(csv.DictReader(open(fs[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0][-28:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Start_Timestamp"]))
       for r in rows if "k_kd_level" in r["Kernel_Name"] or "k_kd_finish" in r["Kernel_Name"]]
runs, cur = [], []
for s in seq:
    cur.append(s)
    if "finish" in s[0]:
        runs.append(cur); cur = []
want = sys.argv[2] if len(sys.argv) > 2 else "<1>"      # the systolic form of the chains by default ("<0>": the v_readlane form)
runs = [b for b in runs if want in b[0][0]]
best = max(runs, key=lambda b: sum(s[1] for s in b))
for i, s in enumerate(best):
    print("%-30s %9.1f us  %6d workgroups  gap before %.1f us" % (s[0], s[1], s[2], ((s[3] - best[i - 1][3]) / 1e3 - best[i - 1][1]) if i else 0.0))
print("kernels %.1f us, span %.1f us" % (sum(s[1] for s in best), (best[-1][3] - best[0][3]) / 1e3 + best[-1][1]))
