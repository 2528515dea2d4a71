#!/bin/bash
# One line per call: the train step of bench.py (60 timed steps, hipGraph replay) on whatever box of the pool this call lands on, with
# the shader clock the waves of ffn_fwd see (in-kernel probe) - the boxes differ by +-3 % and the difference is their clock.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 400 python bench.py --no-cpu-baseline --no-fp32 --no-torch-ref --no-roofline --steps 60 2>/dev/null | grep -E '^\{' | python -c "
import json, sys
r = json.loads(sys.stdin.read())
c = r['config']['clock_mhz']
print(json.dumps({'ms_per_step': r['ms_per_step'], 'icons_per_s': r['value'], 'sclk_sysfs_median_mhz': c['sclk_sysfs']['median_mhz'],
                  'in_kernel_mhz': c.get('in_kernel'), 'dense_ms': r['config'].get('dense_ms_per_step'),
                  'c4_ms': (r.get('secondary') or {}).get('c4_one_stage_train', {}).get('ms_per_step')}))
" | tee -a gpurun_out/r06_box_survey.log
