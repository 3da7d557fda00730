#!/bin/bash
# per-kernel times of tools/probes/attn_bwd_forms_time.py under rocprofv3, once per library variant: <tag> <variant|default> ...
TAG=${1:-r6}; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for v in "$@"; do
  if [ "$v" = default ]; then unset BMT_LIB_PATH; else export BMT_LIB_PATH=$R/bmt_amd/lib/libbmt_hip_$v.so; fi
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_$v -o p -- python $R/tools/probes/attn_bwd_forms_time.py --reps 5 > $R/gpurun_out/${TAG}_${v}_attnprof_run.log 2>&1); echo "== $v rc=$?"
  t=$(find gpurun_out/prof_${TAG}_$v -name "*kernel_trace.csv" | head -1)
  python tools/prof_summary.py "$t" gpurun_out/${TAG}_${v}_attn_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python tools/probes/attn_bwd_forms_time.py --reps 5 (library variant $v)"
  grep "dkvr\|dq32p\|dkvg8\|calls" gpurun_out/${TAG}_${v}_attn_kernel_stats.csv
  rm -rf gpurun_out/prof_${TAG}_$v
done
