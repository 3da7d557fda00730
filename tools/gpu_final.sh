#!/bin/bash
# one GPU call that re-validates the tree after a csrc change: GPU suite -> PMC passes (traffic record for bench.py) -> bench line ->
# attention old/new check -> rocprofv3 kernel stats -> smoke.   usage: bash tools/gpu_final.sh <tag>
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 170 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gputest.log 2>&1; rc=$?
tail -4 gpurun_out/${TAG}_gputest.log; echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit 1
bash tools/gpu_pmc_bench.sh $TAG 2>&1 | tail -6
cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
timeout 150 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; cat gpurun_out/${TAG}_bench.json
timeout 120 python tools/probes/attn_fwd32_check.py > gpurun_out/${TAG}_attn_fwd32_check.txt 2>&1; echo "attn check rc=$?"; tail -7 gpurun_out/${TAG}_attn_fwd32_check.txt
bash tools/gpu_prof.sh $TAG 6 2>&1 | head -30
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
