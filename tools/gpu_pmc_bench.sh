#!/bin/bash
# HBM traffic of the step's kernels from the memory-side counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE in
# their own rocprofv3 passes (kernel-trace only) over the eagerly issued bench step -> gpurun_out/pmc_traffic.json
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timer > $R/gpurun_out/pmcb_$c.log 2>&1
  echo "pass $c rc=$?"
done
cd $R
python - <<'PY'
import collections, csv, glob, json, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmcb_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
            n = re.sub(r"^void ", "", n)
            m = re.match(r"gemm_bf16_kernel<(\d)", n)
            key = f"gemm_planes_x{m.group(1)}" if m else n.split("(")[0].split("<")[0]
            a = agg[key][c]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
out = {"unit": "bytes per launch", "correction": "FETCH_SIZE (KB) x 1024 x 2 (gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md HBM); WRITE_SIZE (KB) x 1024 uncorrected",
       "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-graph", "kernels": {}}
for k, d in agg.items():
    n = max(d["FETCH_SIZE"][0], d["WRITE_SIZE"][0])
    if n == 0:
        continue
    fetch = d["FETCH_SIZE"][1] / max(1, d["FETCH_SIZE"][0]) * 1024 * 2
    write = d["WRITE_SIZE"][1] / max(1, d["WRITE_SIZE"][0]) * 1024
    out["kernels"][k] = {"launches_seen": n, "fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes"] * kv[1]["launches_seen"])[:12]:
    print(f"{k:32s} n={v['launches_seen']:5d} fetch {v['fetch_bytes'] / 1e6:8.2f} MB  write {v['write_bytes'] / 1e6:8.2f} MB per launch")
PY
rm -rf gpurun_out/pmcb_FETCH_SIZE gpurun_out/pmcb_WRITE_SIZE
