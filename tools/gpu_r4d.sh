#!/bin/bash
# round 4, visit d: decoder / generator GEMM operand policy A/B (error + speed), train_prop kernel statistics
TAG=${1:-r04_d}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for e in "BMT_NOP=0" "BMT_DEC_GEMM=w2" "BMT_DEC_GEMM=w2 BMT_X_GEMM=w2"; do
  env $e timeout 300 python -m pytest tests/test_gpu_model.py -q -s -k "seeded_captioning or full_length or deep_config or ten" 2>&1 | grep -E "max \|dlogp\||passed|failed" | sed "s/^/$e: /"
done
bash tools/gpu_ab.sh "BMT_NOP=0" "BMT_DEC_GEMM=w2" "BMT_DEC_GEMM=w2 BMT_X_GEMM=w2" 2>&1 | grep -v amdgpu.ids
bash tools/gpu_prof.sh ${TAG}_prop 3 train_prop 2>&1 | head -45
