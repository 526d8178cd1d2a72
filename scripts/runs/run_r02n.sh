#!/bin/bash
source scripts/gpu_steps.sh
step 900 cfg4_n1.json python bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline --no-hbm-multiply --multiply-reps 3
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/cfg4_n1.json") if l.startswith("{")][-1])
print({k:v for k,v in d.items() if k not in ("kernels","roofline_kernels")})
PY
tail -5 gpurun_out/cfg4_n1.json | cut -c1-400
