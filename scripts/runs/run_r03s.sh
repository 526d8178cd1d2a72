#!/bin/bash
source scripts/gpu_steps.sh
step 600 r03s_mixed.log python -m pytest tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q
tail -3 gpurun_out/r03s_mixed.log
step 300 r03s_bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-multiply
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r03s_bench.json') if l.startswith('{')][-1]); print(d['ms_per_step'], d['mixed_precision'])"
