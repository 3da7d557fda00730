#!/bin/bash
# which ingredient of the batch-in-parts step breaks hipGraph capture: the staged probe on a tiny model under each switch, one process each
# (profiles/r04_i_parts_bisect.txt also has two arms of a since-removed switch that skipped the record_stream calls: no difference)
TAG=${1:-bisect}
mkdir -p gpurun_out
L=gpurun_out/${TAG}_bisect.txt; : > $L
for e in "X=0" "BMT_PARTS_STAGGER=0" "BMT_PARTS_SIDE=0" "BMT_PARTS_STAGGER=0 BMT_PARTS_SIDE=0" "BMT_ENC_STREAMS=1" "PARTS_DEBUG_NO_EAGER=1" "BMT_LN_EMIT=0" "BMT_FUSE_GEN_LOSS=0"; do
  echo "=== $e" >> $L
  env $e timeout 120 python tools/probes/parts_debug.py 2 4 0.0 tiny > /tmp/pd.log 2>&1; rc=$?
  echo "rc=$rc" >> $L
  grep -v "amdgpu.ids\|Extension modules\|^$" /tmp/pd.log | tail -12 | cut -c1-200 >> $L
done
cat $L
timeout 600 python -m pytest tests/test_gpu_parts.py -q -m gpu -s -k "not captured" 2>&1 | grep -v "^$" | tail -60 | cut -c1-250 | tee gpurun_out/${TAG}_parts_tests.log
