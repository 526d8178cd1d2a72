#!/bin/bash
# column sums with 32 records in flight per lane (kernels of their own): parity + bit-identity tests, config 3, P2, small systems
source scripts/gpu_steps.sh
step 900 pytest_r03ag.log python -m pytest tests/test_gpu_hash_mode.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_mixed.py tests/test_gpu_ranks.py -q -x
tail -3 gpurun_out/pytest_r03ag.log
python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03ag_c3.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03ag_c3.json"))
print("config 3: ms_per_step", d["ms_per_step"], "value", d["value"], "iteration", d["roofline_iteration"]["ms_per_iteration"])
print({k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-mixed 2>/dev/null | tail -1 > gpurun_out/r03ag_p2.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03ag_p2.json"))
print("P2: ms_per_step", d["ms_per_step"], "value", d["value"])
print({k: v["avg_ms"] for k, v in d["kernels"].items()})
PY
python scripts/small_latency.py 2>&1 | grep -v amdgpu | tail -6
