#!/bin/bash
# kernel-by-kernel trace of ONE eagerly issued step (dispatch order, durations) -> gpurun_out/trace_step.csv
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -o step -- python $R/bench.py --steps 2 --warmup 2 --no-graph --no-cpu-baseline --no-kernel-timer --no-clock-probe > $R/gpurun_out/trace_run.log 2>&1; echo rc=$?
cd $R
f=$(find gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: everything after the last rng_advance_kernel
idx = [i for i, r in enumerate(rows) if "rng_advance" in r["Kernel_Name"]]
last = rows[idx[-1]:] if idx else rows
t0 = int(last[0]["Start_Timestamp"])
with open("gpurun_out/trace_step.csv", "w") as f:
    f.write("start_us,dur_us,gap_us,kernel\n")
    prev_end = t0
    for r in last:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        f.write(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.2f},{(s - prev_end) / 1e3:.2f},{r['Kernel_Name'][:90]}\n")
        prev_end = max(prev_end, e)
print("kernels in last step:", len(last), "span ms:", (int(last[-1]["End_Timestamp"]) - t0) / 1e6)
PY
rm -rf gpurun_out/trace
