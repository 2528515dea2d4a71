#!/bin/bash
# Build libdsvg_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=../_lib
mkdir -p "$OUT" obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value"
pids=()
for f in gemm gemm_bf16 gemm_bf16_glds layernorm attention attention_mfma embed loss optim assemble match long_seq ffn_fused attn_fused attn_bwd_dx group_stage head_fused pack_images; do
  if [ ! -f obj/$f.o ] || [ $f.hip -nt obj/$f.o ] || [ dsvg_common.h -nt obj/$f.o ] || [ gemm_common.h -nt obj/$f.o ] || [ gemm_bf16.h -nt obj/$f.o ] || [ fused_common.h -nt obj/$f.o ] || [ pack_images.h -nt obj/$f.o ] || [ ../../include/dsvg.h -nt obj/$f.o ]; then
    hipcc $FLAGS -c $f.hip -o obj/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC obj/*.o -o "$OUT/libdsvg_hip.so"
echo "built $OUT/libdsvg_hip.so"
