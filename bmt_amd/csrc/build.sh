#!/bin/bash
# Builds libbmt_hip.so for gfx950 in-tree (bmt_amd/lib/).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
# BMT_VARIANT=<name> BMT_VARIANT_FLAGS="-D..." builds lib/libbmt_hip_<name>.so with those flags (same-box A/B of a build-time choice through
# BMT_LIB_PATH: tools/gpu_r5.sh ablib); the default build ignores both
OUT=../lib
OBJ=obj${BMT_VARIANT:+_$BMT_VARIANT}
LIBNAME=libbmt_hip${BMT_VARIANT:+_$BMT_VARIANT}.so
mkdir -p "$OUT" "$OUT/$OBJ"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result ${BMT_VARIANT:+$BMT_VARIANT_FLAGS}"
SRCS="runtime gemm_bf16 attention attention_bf16 norm elementwise loss optim proposal postprocess ingest raw_memory rank_attn"
pids=()
for f in $SRCS; do
  if [ ! -f "$OUT/$OBJ/$f.o" ] || [ "$f.hip" -nt "$OUT/$OBJ/$f.o" ] || [ common.h -nt "$OUT/$OBJ/$f.o" ] || [ ../../include/bmt_hip.h -nt "$OUT/$OBJ/$f.o" ]; then
    hipcc $FLAGS -c "$f.hip" -o "$OUT/$OBJ/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
# (an explicit object list: a stale object of a source that no longer exists -- obj/gemm.o once -- must not be linked)
OBJS=""; for f in $SRCS; do OBJS="$OBJS $OUT/$OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/$LIBNAME" $OBJS
echo "built $OUT/$LIBNAME"
