#!/bin/bash
# round 4, late: full GPU suite, then same-box A/B of the split-K finish inside the GEMM kernel and of the LayerNorm plane emission at the decoder's width
TAG=${1:-r04_k}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/${TAG}_gputest.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/${TAG}_gputest.log | cut -c1-300
bash tools/gpu_ab.sh "BMT_SPLITK_FUSED=1 BMT_LN_EMIT_ANY=1" "BMT_SPLITK_FUSED=0 BMT_LN_EMIT_ANY=1" "BMT_SPLITK_FUSED=1 BMT_LN_EMIT_ANY=0" "BMT_SPLITK_FUSED=0 BMT_LN_EMIT_ANY=0" 2>&1 | tee gpurun_out/${TAG}_ab_splitk_lnany.txt
