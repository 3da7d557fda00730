#!/bin/bash
# first GPU call of a round that picks up the experiments of bmt_amd/csrc/exp/ (written, compiled and CPU-emulated without a GPU at the end
# of round 2): parity + timing of the attention backward kernels (dQ, dK/dV one- and two-pass) and of the k-major 256 x 256 GEMM against the
# product paths, then the forward probe with the bench's ragged lengths.   usage: bash tools/gpu_exp_all.sh <tag>   (~2 min on the box)
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 240 python tools/probes/attn_bwd32_check.py > gpurun_out/${TAG}_attn_bwd32_check.txt 2>&1; echo "attn_bwd32 rc=$?"; grep -v amdgpu.ids gpurun_out/${TAG}_attn_bwd32_check.txt | tail -22
timeout 120 python tools/probes/gemm_wide_km_check.py > gpurun_out/${TAG}_gemm_wide_km_check.txt 2>&1; echo "gemm_wide_km rc=$?"; grep -v amdgpu.ids gpurun_out/${TAG}_gemm_wide_km_check.txt | tail -14
timeout 60 python tools/probes/attn_fwd32_check.py --probe --ragged > gpurun_out/${TAG}_attn_fwd32_probe_ragged.txt 2>&1; echo "fwd probe rc=$?"; grep -v amdgpu.ids gpurun_out/${TAG}_attn_fwd32_probe_ragged.txt | tail -16
# the whole model through the experimental backward (ops.ATTN_BWD32): the GPU suite's model / kernel tests with the switch on (two-pass dK/dV)
BMT_ATTN_BWD32=2 timeout 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q > gpurun_out/${TAG}_gputest_bwd32.log 2>&1; echo "suite with BMT_ATTN_BWD32=2 rc=$?"; tail -4 gpurun_out/${TAG}_gputest_bwd32.log
