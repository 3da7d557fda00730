#!/bin/bash
# Builds tools/experiments/libbmt_exp.so ON DEMAND (tools/probes/attn_fwd32_check.py runs this; __graft_entry__.build() does not: the
# product's build check covers the product).  exp_lib.hip = bmt_amd/csrc/attention_bf16.hip (included for its helpers and kernels) + the
# forward-kernel experiment driver in one translation unit; runtime.o for bmt_set_error; -Bsymbolic keeps its duplicate C symbols to
# itself when libbmt_hip.so is loaded in the same process.
# (round 4: the measured-and-rejected experiments of rounds 2-3 -- the 32-query backward kernels, the k-major 256 x 256 GEMM, the first
# form of the split backward -- left the tree; their measurements are in DESIGN.md section 6, their sources in the history up to 7e73a20)
set -e
cd "$(dirname "$0")"
LIB=../../bmt_amd/lib
[ -f "$LIB/obj/runtime.o" ] || bash ../../bmt_amd/csrc/build.sh
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -I../../bmt_amd/csrc"
if [ ! -f exp_lib.o ] || [ exp_lib.hip -nt exp_lib.o ] || [ attn_fwd32.hip -nt exp_lib.o ] || [ ../../bmt_amd/csrc/attention_bf16.hip -nt exp_lib.o ]; then
  hipcc $FLAGS $BMT_EXP_FLAGS -c exp_lib.hip -o exp_lib.o
fi
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o libbmt_exp.so exp_lib.o "$LIB/obj/runtime.o"
echo "built tools/experiments/libbmt_exp.so"
