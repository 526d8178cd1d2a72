#!/bin/bash
# round 3: bench with the mixed-precision solve beside the headline; small-system latency baseline; config 3
source scripts/gpu_steps.sh
step 600 r03f_bench.json python bench.py --steps 5 --warmup 2
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r03f_bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("kernel", "avg_ms", "frac", "frac_model", "traffic_over_moved")})
print("mixed", d["mixed_precision"])
print("multiply", d["roofline_multiply"]["avg_ms"], d["roofline_multiply"]["frac"], "hbm corner", d["roofline_multiply_hbm_bound"]["frac"])
PY
step 300 r03f_small.txt python scripts/small_latency.py
cat gpurun_out/r03f_small.txt
step 300 r03f_cfg3.json python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10 --no-cpu-baseline
tail -c 1500 gpurun_out/r03f_cfg3.json
