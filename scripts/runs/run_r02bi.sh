#!/bin/bash
source scripts/gpu_steps.sh
step 1100 pytest_bi.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_bi.log | tail -6
step 600 bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench.json") if l.startswith("{")][-1])
print("P2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline"]["avg_ms_all_launches"], d["roofline_iteration"]["ms_per_iteration"])
for k,v in d["kernels"].items():
    if v["avg_ms"] > 0.1: print(" ", k, v["avg_ms"], v["avg_ms_first_iteration"])
PY
