#!/usr/bin/env python
"""fold the rocprofv3 --pmc passes of tools/gpu_pmc_bench.sh into gpurun_out/<tag>_pmc_traffic.json: per kernel FAMILY (the keys
bench.py maps its kernel classes to) launches seen, average duration, HBM bytes per launch (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE),
HBM GB/s and the share of SIMD cycles the matrix pipe was busy."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import PMC_FAMILY_OF_KERNEL, csrc_digest  # noqa: E402

tag, proc = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "train_cap")
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for i in (1, 2, 3):
    for f in glob.glob(f"gpurun_out/pmcb_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            key = PMC_FAMILY_OF_KERNEL(r["Kernel_Name"])
            a = agg[key][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    if i == 1:
        for f in glob.glob(f"gpurun_out/pmcb_{i}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                d = dur[PMC_FAMILY_OF_KERNEL(r["Kernel_Name"])]
                d[0] += 1
                d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = {"unit": "bytes per launch", "procedure": proc, "csrc_digest": csrc_digest(),
       "correction": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM); WRITE_SIZE (KB) x 1024 "
                     "uncorrected; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); durations under the counter pass",
       "command": f"rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES> --kernel-trace -- python bench.py "
                  f"--procedure {proc} --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timer --no-clock-probe",
       "kernels": {}}
for k, d in agg.items():
    n = max(d["FETCH_SIZE"][0], d["WRITE_SIZE"][0])
    if n == 0:
        continue
    fetch = d["FETCH_SIZE"][1] / max(1, d["FETCH_SIZE"][0]) * 1024 * 2
    write = d["WRITE_SIZE"][1] / max(1, d["WRITE_SIZE"][0]) * 1024
    e = {"launches_seen": n, "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write}
    if dur[k][0]:
        e["avg_us"] = dur[k][1] / dur[k][0]
        e["hbm_gbs"] = (fetch + write) / e["avg_us"] / 1e3
    if d["GRBM_GUI_ACTIVE"][1] > 0:
        e["mfma_busy"] = d["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (d["GRBM_GUI_ACTIVE"][1] / 8 * 1024)
    out["kernels"][k] = e
json.dump(out, open(f"gpurun_out/{tag}_pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches_seen"])[:16]:
    print(f"{k:28s} n={v['launches_seen']:5d} {v.get('avg_us', 0):8.1f} us  fetch {v['fetch_bytes'] / 1e6:8.2f} MB  write {v['write_bytes'] / 1e6:8.2f} MB"
          f"  {v.get('hbm_gbs', 0):7.0f} GB/s  mfma busy {100 * v.get('mfma_busy', 0):5.1f} %")
