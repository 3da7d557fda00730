#!/bin/bash
# round 4, first GPU visit: new tests first (fail fast), then the whole GPU suite, the driver's bench invocation, train_prop both ways,
# the DP parity harness on one rank.   usage: bash tools/gpu_r4a.sh <tag>
TAG=${1:-r04_a}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 400 python -m pytest tests/test_gpu_round4.py -x -q > gpurun_out/${TAG}_new.log 2>&1; echo "new tests rc=$?"; tail -25 gpurun_out/${TAG}_new.log
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round4.py > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest.log | tail -30
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
tail -6 gpurun_out/${TAG}_bench.err
python tools/bench_summary.py gpurun_out/${TAG}_bench.json 2>/dev/null | head -40
for hc in w2 x3; do
  BMT_HEAD_CONV=$hc timeout 300 python bench.py --procedure train_prop --steps 6 --warmup 2 --no-cpu-baseline --no-clock-probe > gpurun_out/${TAG}_prop_$hc.json 2> gpurun_out/${TAG}_prop_$hc.err; echo "train_prop $hc rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/${TAG}_prop_$hc.json").read().strip().splitlines()[-1])
    print("train_prop $hc: ms/step", d["ms_per_step"], "valid", d.get("roofline", {}).get("valid"))
    for k, v in list(d.get("kernel_classes", {}).items())[:6]:
        print(f"  {k:40s} {v['ms_per_step']:7.3f} ms  {v['tflops']:7.1f} TF  n={v['launches_per_step']:.0f}")
except Exception as e:
    print("parse failed", e)
PY
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29711 tools/dp_parity_2gpu.py > gpurun_out/${TAG}_dp_parity_1rank.log 2>&1; echo "dp parity (1 rank) rc=$?"; grep -E "ok|DIFFER|DP-PARITY|Error" gpurun_out/${TAG}_dp_parity_1rank.log | tail -8
