#!/bin/bash
# one GPU-box visit: gpu tests, the bench line (train_cap + train_prop), optional extras.  usage: tools/gpu_round.sh <tag> [extra shell command ...]
TAG=${1:-x}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_gputest.log
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest.log | tail -40
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/${TAG}_bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("ms/step", d["ms_per_step"], "value", d["value"], "dtype", d["dtype"])
    print("roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "frac_issued", "avg_launch_us", "traffic")})
    print("attention", d.get("attention_roofline"))
    for k, v in d["kernel_classes"].items():
        print(f"  {k:40s} {v['ms_per_step']:7.3f} ms  {v['tflops']:7.1f} TF  n={v['launches_per_step']:.0f}")
    print("cpu", d.get("cpu_baseline"))
except Exception as e:
    print("bench parse failed", e)
PY
for c in "$@"; do echo "+ $c"; eval "$c"; done
