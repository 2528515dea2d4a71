#!/bin/bash
# A/B of environment knobs on ONE box: `scripts/ab.sh "VAR_A=1" "VAR_B=0 VAR_C=2" ...` runs bench.py (train step only, 60 timed
# steps) under every setting in turn, three rounds, and prints ms/step per setting (boxes of the pool differ by +-3 %).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
for round in 1 2 3; do
  for cfg in "$@"; do
    ms=$(env $cfg timeout 300 python bench.py --no-cpu-baseline --no-fp32 --no-torch-ref --no-roofline --no-extra-legs --steps 60 2>/dev/null | grep -E '^\{' | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $round  [$cfg]  $ms ms/step"
  done
done
