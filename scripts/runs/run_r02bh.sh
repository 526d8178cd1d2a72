#!/bin/bash
source scripts/gpu_steps.sh
step 600 pytest_bh.log python -m pytest tests/test_gpu_parity.py tests/test_bench_binary.py tests/test_bench_launcher.py -m gpu -q -x -k "profile or profiling or bench or launcher or tfqmr_mode"
tail -4 gpurun_out/pytest_bh.log
step 600 bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench.json") if l.startswith("{")][-1])
print("P2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline"]["avg_ms_all_launches"])
for k,v in d["kernels"].items(): print(" ", k, v)
PY
