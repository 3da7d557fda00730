#!/bin/bash
# SQ / LDS / MFMA / L2 counters of one kernel configuration (tools/pmc_case.py args), separate rocprofv3 --pmc passes (kernel-trace only)
# usage: tools/gpu_pmc_case.sh <tag> <pmc_case args...>   -> gpurun_out/<tag>_pmc.csv
TAG=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0; dirs=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcc_${TAG}_$i -o p -- python $R/tools/pmc_case.py "$@" > $R/gpurun_out/pmcc_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
  dirs="$dirs $(dirname $(find $R/gpurun_out/pmcc_${TAG}_$i -name '*counter_collection.csv' | head -1))"
done
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_pmc.csv "rocprofv3 --pmc (3 passes) -- python tools/pmc_case.py $*" $dirs | cut -c1-400 | head -30
rm -rf gpurun_out/pmcc_${TAG}_*
