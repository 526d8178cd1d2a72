#!/bin/bash
source scripts/gpu_steps.sh
step 300 pytest_prof.log python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "profiling_levels or plan_reuse"
tail -15 gpurun_out/pytest_prof.log
step 600 bench_r02u.json python bench.py --steps 5 --warmup 2
python - <<'PY'
import json
for line in open("gpurun_out/bench_r02u.json"):
    if line.startswith("{"):
        d = json.loads(line)
        print(d["value"], d["ms_per_step"], d["iterations_per_solve"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline_iteration"])
        for k, v in d["kernels"].items(): print(" ", k, v)
        print(d.get("cpu_baseline"))
PY
