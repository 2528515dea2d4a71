#!/bin/bash
# Same-box A/B of two BUILDS of the library: put them at deepsvg_amd/_lib/libdsvg_hip_old.so / _new.so (git stash; build.sh; cp ...),
# three alternating rounds of bench.py's train step (60 timed steps each).
cd "$(dirname "$0")/.."
for round in 1 2 3; do
  for v in old new; do
    cp deepsvg_amd/_lib/libdsvg_hip_$v.so deepsvg_amd/_lib/libdsvg_hip.so
    ms=$(timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-torch-ref --no-roofline --no-extra-legs --steps 60 2>/dev/null | grep -E '^\{' | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round [$v draws] $ms ms/step"
  done
done
