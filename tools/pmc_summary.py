#!/usr/bin/env python
"""mean counter value per dispatch, per (kernel, workgroups), over the rocprofv3 --pmc passes under the given directories.
usage: pmc_summary.py out.csv "comment" dir1 [dir2 ...]"""
import collections
import csv
import glob
import re
import sys

out, comment, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "anonymous namespace" not in n:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            n = re.sub(r"^void ", "", n).split("(")[0]
            wg = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
            a = agg[(n, wg)][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
names = sorted({c for v in agg.values() for c in v})
with open(out, "w") as fh:
    fh.write("# " + comment + "\n")
    w = csv.writer(fh)
    w.writerow(["kernel", "workgroups", "dispatches"] + names)
    for (n, wg), v in sorted(agg.items()):
        disp = max(x[0] for x in v.values())
        w.writerow([n, wg, disp] + ["%.4g" % (v[c][1] / v[c][0]) if c in v else "" for c in names])
print(open(out).read()[:6000])
