#!/bin/bash
# round 4, visit e: the proposal-path kernels (gate-fused planes, weight re-layout, tiled decode / loss), whole suite, train_prop line + stats
TAG=${1:-r04_e}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_proposal.py -q > gpurun_out/${TAG}_new.log 2>&1; echo "new + proposal tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_new.log | tail -20
BMT_GRAD_REPORT=1 timeout 900 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_round4.py --deselect tests/test_gpu_proposal.py > gpurun_out/${TAG}_gputest.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_gputest.log | tail -30
grep -E "global relative gradient error" gpurun_out/${TAG}_gputest.log | sort -k5 -g | tail -4
grep -E "^ +[0-9.]+%" gpurun_out/${TAG}_gputest.log | sort -rn | head -6
for e in "BMT_NOP=0" "BMT_FUSE_GATE=0 BMT_PROP_TILED=0"; do
  env $e timeout 300 python bench.py --procedure train_prop --steps 10 --warmup 3 --no-cpu-baseline --no-clock-probe > gpurun_out/${TAG}_prop.json 2> gpurun_out/${TAG}_prop.err; echo "train_prop [$e] rc=$?"
  python tools/bench_summary.py gpurun_out/${TAG}_prop.json 2>/dev/null | head -9
  [ "$e" = "BMT_NOP=0" ] && cp gpurun_out/${TAG}_prop.json gpurun_out/${TAG}_bench_train_prop.json
done
bash tools/gpu_prof.sh ${TAG}_prop 3 train_prop 2>&1 | head -32
