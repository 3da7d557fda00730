#!/usr/bin/env python
"""per-kernel means of the rocprofv3 --pmc passes of tools/gpu_attnpmc.sh: usage attn_pmc_summary.py <tag> <npass>"""
import collections
import csv
import glob
import sys

tag, npass = sys.argv[1], int(sys.argv[2])
lines = []
for i in range(1, npass + 1):
    fs = glob.glob(f"gpurun_out/pmca_{i}/**/*counter_collection.csv", recursive=True) + glob.glob(f"gpurun_out/pmca_{i}/*counter_collection.csv")
    if not fs:
        lines.append(f"pass {i}: no counter csv")
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "attn" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k in sorted(agg):
        lines.append(f"{k}  launches {max(n[(k, c)] for c in agg[k])}  " + "  ".join(f"{c}={v / n[(k, c)]:.4g}" for c, v in sorted(agg[k].items())))
open(f"gpurun_out/{tag}_attn_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
