#!/bin/bash
source scripts/gpu_steps.sh
step 900 c5big.json python bench.py --workload st:8:8:z:512:512:8 --steps 3 --warmup 1 --no-cpu-baseline --no-hbm-multiply
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/c5big.json") if l.startswith("{")][-1])
print(d["config"]["workload"], d["value"], d["ms_per_step"], d["iterations_per_solve"], d["solve_status"], d["residual"], d["buffer_GB_per_gpu"])
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline_multiply"]["frac"], d["roofline_iteration"])
print(d["roofline_kernels"])
PY
