#!/bin/bash
# kernel-level times (rocprofv3 --kernel-trace) of tools/probes/gemm_k128_time.py under a few settings: the python-side loop issues a launch
# every ~20 us, so event timing of back-to-back launches cannot resolve kernels shorter than that
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in ${K128_CFGS:-"BMT_GEMM_K128=0" "BMT_GEMM_K128=1" "BMT_K128_DBG=1" "BMT_K128_DBG=3"}; do
  rm -rf /tmp/k128prof
  (cd /tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/k128prof -o p -- python $R/tools/probes/gemm_k128_time.py > /tmp/k128run.log 2>&1)
  t=$(find /tmp/k128prof -name "*kernel_trace.csv" | head -1)
  echo "== $cfg"
  python - "$t" <<'EOP'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# launches in issue order; group consecutive runs of the GEMM kernels (each shape is timed as 3 + 5 x 20 launches)
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]) if "Grid_Size_X" in r else 0) for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
       if "gemm" in r["Kernel_Name"] and "planes" not in r["Kernel_Name"]]
runs = []
for name, us, g in seq:
    key = (name.split("(")[0][:60].replace("void (anonymous namespace)::", ""), g)
    if runs and runs[-1][0] == key:
        runs[-1][1].append(us)
    else:
        runs.append([key, [us]])
for key, v in runs:
    if len(v) >= 50:
        v = sorted(v)
        print(f"  {key[0]:62s} grid {key[1]:7d}  n={len(v):4d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
EOP
done
