#!/bin/bash
# one GPU call that re-validates the tree after a csrc change: GPU suite -> PMC passes (traffic record for bench.py) -> bench line (the
# driver's invocation) -> train_prop line -> rocprofv3 kernel stats -> graph-replay timeline -> smoke.   usage: bash tools/gpu_final.sh <tag>
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 400 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1; rc=$?
tail -4 gpurun_out/${TAG}_gputest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
bash tools/gpu_pmc_bench.sh $TAG 2>&1 | tail -6
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
grep "bench\]" gpurun_out/${TAG}_bench.err | tail -5
python tools/bench_summary.py gpurun_out/${TAG}_bench.json
# train_prop has a PMC record of its own (bench.py takes the newest record taken over the procedure it runs)
bash tools/gpu_pmc_bench.sh ${TAG}_prop train_prop 2>&1 | grep -i "conv\|pass"
cp gpurun_out/${TAG}_prop_pmc_traffic.json profiles/${TAG}_prop_pmc_traffic.json
timeout 300 python bench.py --procedure train_prop --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_train_prop.json 2> gpurun_out/${TAG}_prop.err; echo "train_prop rc=$?"
python tools/bench_summary.py gpurun_out/${TAG}_bench_train_prop.json | head -10
bash tools/gpu_prof.sh $TAG 6 2>&1 | head -40
bash tools/gpu_prof.sh ${TAG}_prop 6 train_prop 2>&1 | head -30
bash tools/gpu_timeline.sh $TAG 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
# the rows either side of the path and the deep configuration, with the same tree (numbers for DESIGN.md section 6)
timeout 200 python tools/deep_config_step.py 2>&1 | grep -v amdgpu | tail -1 | tee gpurun_out/${TAG}_deep_config_step.txt
timeout 120 python tools/decode_bench.py > gpurun_out/${TAG}_decode_bench.json 2>/dev/null; tail -c 600 gpurun_out/${TAG}_decode_bench.json; echo
timeout 120 python tools/postprocess_bench.py > gpurun_out/${TAG}_postprocess_bench.json 2>/dev/null; tail -c 400 gpurun_out/${TAG}_postprocess_bench.json; echo
timeout 120 python tools/ingest_bench.py > gpurun_out/${TAG}_ingest_bench.json 2>/dev/null; tail -c 400 gpurun_out/${TAG}_ingest_bench.json; echo
