#!/bin/bash
# SQ / LDS / MFMA counters of the split attention backward kernels (and the product's, same process): separate rocprofv3 --pmc passes
# usage: tools/gpu_pmc_split.sh <tag> [--shape B,H,Sq,Sk,dk]   -> gpurun_out/<tag>_pmc.csv
TAG=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0; dirs=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_${TAG}_$i -o p -- python $R/tools/probes/attn_bwd_split_check.py --pmc-case "$@" > $R/gpurun_out/pmcs_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"; tail -2 $R/gpurun_out/pmcs_${TAG}_$i.log | cut -c1-300
  f=$(find $R/gpurun_out/pmcs_${TAG}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && dirs="$dirs $(dirname $f)"
done
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_pmc.csv "rocprofv3 --pmc (3 passes) -- python tools/probes/attn_bwd_split_check.py --pmc-case $*" $dirs | grep -i "kernel,\|attn_bwd\|attn_delta" | cut -c1-700
rm -rf gpurun_out/pmcs_${TAG}_*
