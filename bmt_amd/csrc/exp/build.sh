#!/bin/bash
# Builds the experiment library bmt_amd/lib/libbmt_exp.so (NOT loaded by the product; tools/probes/attn_fwd32_check.py opens it).
# exp_lib.hip = attention_bf16.hip (included for its helpers and kernels) + the attention experiment drivers in one translation unit;
# gemm_wide_km.hip = gemm_bf16.hip + the k-major 256 x 256 kernel in another; runtime.o for
# bmt_set_error; -Bsymbolic keeps its duplicate C symbols to itself when libbmt_hip.so is loaded in the same process.
set -e
cd "$(dirname "$0")"
OUT=../../lib
[ -f "$OUT/obj/runtime.o" ] || bash ../build.sh
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result"
mkdir -p "$OUT/obj_exp"
# (recompiled only when a source is newer than its object: the product files they include count as sources)
newer() { local o=$1; shift; [ ! -f "$o" ] && return 0; for s in "$@"; do [ "$s" -nt "$o" ] && return 0; done; return 1; }
pids=()
if newer "$OUT/obj_exp/exp_lib.o" exp_lib.hip attn_fwd32.hip attn_bwd32.hip attn_bwd_split.hip ../attention_bf16.hip ../common.h ../../../include/bmt_hip.h; then
  hipcc $FLAGS $BMT_EXP_FLAGS -c exp_lib.hip -o "$OUT/obj_exp/exp_lib.o" & pids+=($!)
fi
if newer "$OUT/obj_exp/gemm_wide_km.o" gemm_wide_km.hip ../gemm_bf16.hip ../common.h ../../../include/bmt_hip.h; then
  hipcc $FLAGS $BMT_EXP_FLAGS -c gemm_wide_km.hip -o "$OUT/obj_exp/gemm_wide_km.o" & pids+=($!)
fi
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o "$OUT/libbmt_exp.so" "$OUT/obj_exp/exp_lib.o" "$OUT/obj_exp/gemm_wide_km.o" "$OUT/obj/runtime.o"
echo "built $OUT/libbmt_exp.so"
