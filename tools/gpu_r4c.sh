#!/bin/bash
# round 4, visit c: decoder fan-in / fan-out kernels (model tests), FFN-1 on one fp16 plane: error on every captioning fixture + same-box A/B
TAG=${1:-r04_c}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_unimodal.py tests/test_gpu_dp.py tests/test_gpu_round4.py -q > gpurun_out/${TAG}_model.log 2>&1; echo "model tests rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/${TAG}_model.log | tail -20
for v in 0 1; do
  BMT_FFN1_ONE_PLANE=$v timeout 300 python -m pytest tests/test_gpu_model.py -q -s -k "seeded_captioning or full_length or deep_config or ten" 2>&1 | grep -E "max \|dlogp\||passed|failed" | sed "s/^/FFN1_ONE_PLANE=$v: /"
done
bash tools/gpu_ab.sh "BMT_FFN1_ONE_PLANE=0" "BMT_FFN1_ONE_PLANE=1" 2>&1 | grep -v amdgpu.ids
bash tools/gpu_prof.sh $TAG 4 2>&1 | grep -i "at::native\|rocclr\|total kernel\|cat2\|add_kernel\|split" | head
