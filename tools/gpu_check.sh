#!/bin/bash
# One gpurun call: every -m gpu test group in its own process (a device fault in one group must not mask the others),
# logs under gpurun_out/.  Usage: tools/gpu_check.sh [extra pytest args]
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { # name, timeout, pytest selection...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.txt
  timeout $t python -m pytest "$@" -q -m gpu -p no:cacheprovider > gpurun_out/$name.log 2>&1
  echo "rc=$? $(tail -1 gpurun_out/$name.log)" | tee -a gpurun_out/summary.txt
}
rm -f gpurun_out/summary.txt
run gemm 300 tests/test_gpu_kernels.py -k "gemm"
run attn_fwd 300 tests/test_gpu_kernels.py -k "attention_forward or fully_masked or plane_outputs"
run attn_bwd 300 tests/test_gpu_kernels.py -k "attention_backward or attention_dropout"
run misc 300 tests/test_gpu_kernels.py -k "not gemm and not attention"
run model 600 tests/test_gpu_model.py
run proposal 300 tests/test_gpu_proposal.py
cat gpurun_out/summary.txt
