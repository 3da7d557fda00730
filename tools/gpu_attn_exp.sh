#!/bin/bash
# one GPU call for the experimental attention forward: parity + timing, then one PMC pass over the A-self shape (old and new kernel)
# usage (on the GPU box, via gpurun): bash tools/gpu_attn_exp.sh   -> gpurun_out/attn_fwd32_check.txt, gpurun_out/attn_fwd32_pmc.csv
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 200 python tools/probes/attn_fwd32_check.py > gpurun_out/attn_fwd32_check.txt 2>&1
echo "check rc=$?"; tail -30 gpurun_out/attn_fwd32_check.txt
cd /tmp
timeout 100 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn32 -o p -- python $R/tools/probes/attn_fwd32_check.py --pmc-case > $R/gpurun_out/pmc_attn32.log 2>&1
echo "pmc rc=$?"
cd $R
f=$(find gpurun_out/pmc_attn32 -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python tools/pmc_summary.py gpurun_out/attn_fwd32_pmc.csv "rocprofv3 --pmc -- attn_fwd32_check.py --pmc-case" $(dirname $f) | grep -i "attn_fwd" | cut -c1-600
rm -rf gpurun_out/pmc_attn32
