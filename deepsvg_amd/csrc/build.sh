#!/bin/bash
# Build libdsvg_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
#   build.sh            incremental: a source is recompiled when the SHA-256 of (its text + every shared header + the flags)
#                       differs from the one its object was built from (obj/<name>.sha) - not by time stamps, which a
#                       checkout or a copy of the tree rewrites
#   build.sh --clean    drop every object first: a full compile from the sources (what __graft_entry__.build() runs, ~80 s)
set -e
cd "$(dirname "$0")"
OUT=../_lib
if [ "$1" = "--clean" ]; then rm -rf obj "$OUT/libdsvg_hip.so"; fi
mkdir -p "$OUT" obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value"
HDRS="dsvg_common.h gemm_common.h gemm_bf16.h fused_common.h pack_images.h ../../include/dsvg.h"
HSHA=$( (echo "$FLAGS"; cat $HDRS) | sha256sum | cut -d' ' -f1)
SRCS="gemm gemm_bf16 gemm_bf16_glds layernorm attention attention_mfma embed loss optim assemble match long_seq ffn_fused attn_fused attn_bwd_dx group_stage head_fused pack_images"
pids=()
names=()
for f in $SRCS; do
  want=$( (echo "$HSHA"; cat $f.hip) | sha256sum | cut -d' ' -f1)
  have=$(cat obj/$f.sha 2>/dev/null || true)
  if [ ! -f obj/$f.o ] || [ "$want" != "$have" ]; then
    rm -f obj/$f.sha
    ( hipcc $FLAGS -c $f.hip -o obj/$f.o && echo "$want" > obj/$f.sha ) &
    pids+=($!)
    names+=($f)
  fi
done
for p in "${pids[@]}"; do wait $p; done
echo "compiled ${#names[@]} of $(echo $SRCS | wc -w) sources: ${names[*]}"
objs=""
for f in $SRCS; do objs="$objs obj/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT/libdsvg_hip.so"
echo "built $OUT/libdsvg_hip.so"
