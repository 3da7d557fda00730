#!/bin/bash
# Builds libbmt_hip.so for gfx950 in-tree (bmt_amd/lib/).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../lib
mkdir -p "$OUT" "$OUT/obj"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result"
SRCS="runtime gemm_bf16 attention attention_bf16 norm elementwise loss optim proposal postprocess ingest"
pids=()
for f in $SRCS; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$f.hip" -nt "$OUT/obj/$f.o" ] || [ common.h -nt "$OUT/obj/$f.o" ] || [ ../../include/bmt_hip.h -nt "$OUT/obj/$f.o" ]; then
    hipcc $FLAGS -c "$f.hip" -o "$OUT/obj/$f.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
# (an explicit object list: a stale object of a source that no longer exists -- obj/gemm.o once -- must not be linked)
OBJS=""; for f in $SRCS; do OBJS="$OBJS $OUT/obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libbmt_hip.so" $OBJS
echo "built $OUT/libbmt_hip.so"
