#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02b.log python -m pytest tests -m gpu -q -x
step 400 bench_r02b.json python bench.py --steps 5 --warmup 2
step 300 bench_r02b_ilv0.json env TFQMRGPU_ILV=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline
tail -5 gpurun_out/pytest_r02b.log
python - <<'PY'
import json
for f in ("bench_r02b.json","bench_r02b_ilv0.json"):
    try:
        d=json.loads([l for l in open("gpurun_out/"+f) if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_ms"], d["roofline"]["frac"], {k:v["avg_ms"] for k,v in d["kernels"].items()})
        print("   cpu", d.get("cpu_baseline"))
    except Exception as e: print(f, "failed", e)
PY
