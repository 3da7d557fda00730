#!/bin/bash
# SQ / LDS / MFMA / TCC counters of the reduction-of-128 GEMM on one shape (N = $2, default 3072), 3 + 3 separate rocprofv3 --pmc passes
# usage: tools/gpu_pmc_k128.sh <tag> [N] [ENV=VALUE ...]   -> gpurun_out/<tag>_k128_pmc.csv
TAG=$1; N=${2:-3072}; shift; shift
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0; dirs=""
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_WRREQ_STALL_sum"; do
  i=$((i+1))
  env "$@" timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmck_${TAG}_$i -o p -- python $R/tools/probes/gemm_k128_time.py --one $N > $R/gpurun_out/pmck_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
  f=$(find $R/gpurun_out/pmck_${TAG}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && dirs="$dirs $(dirname $f)"
done
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_k128_pmc.csv "rocprofv3 --pmc (6 passes) -- $* python tools/probes/gemm_k128_time.py --one $N" $dirs | grep -i "kernel,\|gemm_" | cut -c1-900
rm -rf gpurun_out/pmck_${TAG}_*
