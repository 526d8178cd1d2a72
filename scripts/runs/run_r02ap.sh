#!/bin/bash
source scripts/gpu_steps.sh
step 600 bench.json python bench.py --steps 5 --warmup 2
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench.json") if l.startswith("{")][-1])
print("P2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"])
for k in ("roofline_multiply","roofline_multiply_native_api","roofline_multiply_hbm_bound"): print(k, d[k]["avg_ms"], d[k]["achieved"], d[k]["frac"])
PY
