#!/bin/bash
# HBM-side traffic of the split attention backward kernels: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), L2 hit / miss in a third
TAG=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
i=0; dirs=""
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmct_${TAG}_$i -o p -- python $R/tools/probes/attn_bwd_split_check.py --pmc-case "$@" > $R/gpurun_out/pmct_${TAG}_$i.log 2>&1
  echo "pmc pass $i rc=$?"
  f=$(find $R/gpurun_out/pmct_${TAG}_$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && dirs="$dirs $(dirname $f)"
done
cd $R
python tools/pmc_summary.py gpurun_out/${TAG}_traffic.csv "rocprofv3 --pmc (FETCH_SIZE | WRITE_SIZE | TCC hit/miss: 3 passes; FETCH_SIZE is in KB and counts half of a wide streaming read on gfx950) -- python tools/probes/attn_bwd_split_check.py --pmc-case $*" $dirs | grep -i "kernel,\|attn_bwd\|attn_delta" | cut -c1-300
rm -rf gpurun_out/pmct_${TAG}_*
