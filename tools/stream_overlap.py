"""Does the streamed step overlap?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --stream` and reports, per k_align launch, how long the NEXT batch's pre-kernels
(k_preprocess_scans, k_cull_estimate: queued on the context's second stream while the launch is in flight) ran INSIDE it.
usage: python tools/stream_overlap.py gpurun_out/<tag>/stream_trace  ->  a table on stdout"""
import csv
import glob
import os
import sys

root = sys.argv[1]
rows = []
for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
aligns = [r for r in rows if "k_align" in r[2]]
pres = [r for r in rows if "k_preprocess_scans" in r[2] or "k_cull_estimate" in r[2]]
if len(aligns) < 10:
    raise SystemExit("too few k_align launches in the trace")
# the streamed steps: launches with a k_preprocess_scans launch since the previous one (bench.py --stream ends with a loop of resident-input steps: not those),
# and of them the second half (steady state)
prep_starts = sorted(p0 for p0, _, name, _ in pres if "k_preprocess_scans" in name)
import bisect
streamed = [a for k, a in enumerate(aligns) if k > 0 and bisect.bisect_left(prep_starts, aligns[k - 1][0]) < bisect.bisect_left(prep_starts, a[0])]
if len(streamed) >= 10:
    aligns = streamed
aligns = aligns[len(aligns) // 2:]
t_first = aligns[0][0]
tot_in, tot_pre, n_inside, gaps = 0, 0, 0, []
for i, (a0, a1, _, _) in enumerate(aligns):
    for (p0, p1, name, _) in pres:
        if p1 <= a0 or p0 >= a1:
            continue
        ov = min(a1, p1) - max(a0, p0)
        tot_in += ov; n_inside += 1
    if i + 1 < len(aligns):
        gaps.append(aligns[i + 1][0] - a1)
for (p0, p1, name, _) in pres:
    if p0 >= t_first:
        tot_pre += p1 - p0
dur = [a1 - a0 for a0, a1, _, _ in aligns]
print("k_align launches (second half of the trace): %d, mean %.1f us" % (len(aligns), sum(dur) / len(dur) / 1e3))
print("pre-kernels of the NEXT batch running inside a k_align launch: %d launches, %.1f us per step of their %.1f us per step" % (
    n_inside, tot_in / len(aligns) / 1e3, tot_pre / len(aligns) / 1e3))
print("gap between the end of one k_align and the start of the next: mean %.1f us, median %.1f us" % (sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
print("step = launch + gap: %.1f us" % ((sum(dur) / len(dur) + sum(gaps) / len(gaps)) / 1e3))
# (since the lane streams two k_align launches OVERLAP: a launch's own duration and the end-to-start "gap" say little -- the step is the start-to-start interval)
starts = [a0 for a0, _, _, _ in aligns]
iv = sorted(starts[i + 1] - starts[i] for i in range(len(starts) - 1))
both = sum(max(0, min(aligns[i][1], aligns[i + 1][1]) - aligns[i + 1][0]) for i in range(len(aligns) - 1))
print("start of one k_align to the start of the next: mean %.1f us, median %.1f us; two launches on the chip together %.0f %% of the time" % (
    sum(iv) / len(iv) / 1e3, iv[len(iv) // 2] / 1e3, 100.0 * both / max(1, aligns[-1][1] - aligns[0][0])))
g = sorted(gaps)
print("gap percentiles [us]: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (g[len(g) // 10] / 1e3, g[len(g) // 2] / 1e3, g[len(g) * 9 // 10] / 1e3, g[-1] / 1e3))
# three steps of the steady state, every launch on every queue (and the copies, if the trace has them)
copies = []
for path in sorted(glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?"), "-"))
k = len(aligns) // 2
w0, w1 = aligns[k][0] - 50000, aligns[min(k + 3, len(aligns) - 1)][1]
for (s0, s1, name, q) in sorted(rows + copies):
    if s0 >= w0 and s0 <= w1:
        print("  %10.1f us  -> %10.1f  (%8.1f us)  q=%s  %s" % ((s0 - w0) / 1e3, (s1 - w0) / 1e3, (s1 - s0) / 1e3, q, name[:60]))
