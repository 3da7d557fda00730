#!/bin/bash
# HBM traffic and matrix-pipe occupancy of the step's kernels: FETCH_SIZE, WRITE_SIZE and the SQ / GRBM cycle counters in their own
# rocprofv3 --pmc passes (kernel-trace only; MI355X_MICROARCH.md, HBM section) over the eagerly issued bench step
#   -> gpurun_out/<tag>_pmc_traffic.json   (copy to profiles/: bench.py reads the newest profiles/*pmc_traffic.json whose csrc_digest
#      matches the tree).   usage: tools/gpu_pmc_bench.sh <tag> [train_cap|train_prop]
TAG=${1:-x}; PROC=${2:-train_cap}
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
export BMT_ENC_STREAMS=1   # kernels one at a time: isolated durations / counters (the bench's timed region forks two streams)
cd /tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$i -o p -- python $R/bench.py --procedure $PROC --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-timer --no-clock-probe > $R/gpurun_out/pmcb_$i.log 2>&1
  echo "pass $i ($c) rc=$?"
done
cd $R
python tools/pmc_traffic.py $TAG $PROC
rm -rf gpurun_out/pmcb_1 gpurun_out/pmcb_2 gpurun_out/pmcb_3
