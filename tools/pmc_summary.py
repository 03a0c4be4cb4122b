"""Summarise rocprofv3 --pmc CSVs (gpurun_out/prof/<pass>/**/_counter_collection.csv) per kernel.
usage: python tools/pmc_summary.py gpurun_out/prof [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
want = sys.argv[2] if len(sys.argv) > 2 else "k_align"
acc = defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per_dispatch = defaultdict(dict)
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]      # ("k_align" means the k_align<...> template itself, not k_align_seq / _narrow / _pair: round 6)
        if not ((want + "<" in name or want + "I" in name) if want == "k_align" else want in name):
            continue
        per_dispatch[row["Dispatch_Id"]][row["Counter_Name"]] = float(row["Counter_Value"])
    for d in per_dispatch.values():
        for k, v in d.items():
            acc[k].append(v)
print("counter,launches,mean_per_launch")
for k in sorted(acc):
    v = acc[k]
    print("%s,%d,%.6g" % (k, len(v), sum(v) / len(v)))
