#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) into a per-kernel stats CSV:
name, calls, total_ms, avg_us, min_us, max_us, pct.   usage: prof_summary.py <in.db|in.csv> <out.csv> [note] [steps last_step.csv]
With ``steps`` (the number of identical eager steps in the run) and a csv input, the dispatches of the LAST step are also written in
issue order: kernel, grid, workgroup, duration us, idle gap before it us."""
import csv
import re
import sqlite3
import sys


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    return cur.execute("select name, end-start from kernels").fetchall()


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    agg = {}
    for name, d in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name)
        a = agg.setdefault(name, [0, 0, 10**18, 0])
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    with open(dst, "w", newline="") as f:
        if note:
            f.write(f"# {note}\n")
        f.write(f"# total kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches\n")
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"])
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([name[:160], a[0], f"{a[1] / 1e6:.3f}", f"{a[1] / a[0] / 1e3:.2f}", f"{a[2] / 1e3:.2f}", f"{a[3] / 1e3:.2f}",
                        f"{100 * a[1] / tot:.2f}"])


def last_step(src, steps, dst):
    with open(src) as f:
        rows = list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # one optimizer step = what lies between two launches of the Adam kernel (the dispatch count divided by the number of steps also
    # counts the first step's one-off launches: weight-plane registration, optimizer state)
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    if len(adam) >= 2:
        rows = rows[adam[-2] + 1:adam[-1] + 1]
    else:
        rows = rows[-(len(rows) // steps):]
    n = len(rows)
    with open(dst, "w", newline="") as f:
        f.write(f"# last of {steps} eager steps: {n} dispatches, {(int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6:.3f} ms from first start to last end\n")
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "workgroup", "us", "gap_us"])
        prev = None
        for r in rows:
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            name = re.sub(r"^void ", "", name).split("(")[0][:70]
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gx = r.get("Grid_Size_X", r.get("Grid_Size", ""))
            try:                       # threads of the whole grid (x * y * z): batched launches put the batch in y
                gx = int(gx) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
            except ValueError:
                pass
            w.writerow([name, gx, r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")),
                        f"{(en - st) / 1e3:.2f}", f"{(st - prev) / 1e3:.2f}" if prev is not None else ""])
            prev = en


if __name__ == "__main__":
    if len(sys.argv) > 5 and sys.argv[1].endswith(".csv"):
        last_step(sys.argv[1], int(sys.argv[4]), sys.argv[5])
    main()
