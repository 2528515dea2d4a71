#!/bin/bash
# gfx950 assembly of every HIP source of the library (device side only) into DIR (default /tmp/dsvg_isa): the input of
# scripts/isa_scan.py, isa_loop_mix.py and isa_diff.py.  hipcc cross-compiles; no GPU needed.
DIR=${1:-/tmp/dsvg_isa}
mkdir -p "$DIR"
cd "$(dirname "$0")/../deepsvg_amd/csrc"
for f in *.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -S --cuda-device-only \
        "$f" -o "$DIR/${f%.hip}.s" 2>/dev/null &
done
wait
ls "$DIR"/*.s | wc -l
