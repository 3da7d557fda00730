#!/bin/bash
# SQ / LDS / traffic counters of the attention backward kernels (tools/probes/attn_bwd_forms_time.py --reps 1): <tag>
TAG=${1:-r6}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmca_$i -o p -- python $R/tools/probes/attn_bwd_forms_time.py --reps 1 > $R/gpurun_out/pmca_$i.log 2>&1
  echo "pass $i ($c) rc=$?"
done
cd $R
python tools/attn_pmc_summary.py $TAG 4 2>&1
for i in 1 2 3 4; do head -c 3000 gpurun_out/pmca_$i.log > gpurun_out/${TAG}_pmca_$i.tail; wc -l gpurun_out/pmca_$i/*.csv >> gpurun_out/${TAG}_pmca_$i.tail; head -3 gpurun_out/pmca_$i/p_counter_collection.csv >> gpurun_out/${TAG}_pmca_$i.tail; find gpurun_out/pmca_$i | head -5 >> gpurun_out/${TAG}_pmca_$i.tail; done; rm -rf gpurun_out/pmca_*
