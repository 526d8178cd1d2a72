#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do step 300 c3_lb3_$r.txt python scripts/bench_multiply.py stencil3d_32x32_c 20; done
for f in gpurun_out/c3_lb3_*.txt; do echo "== $f"; grep -E "spmm|per iter|status" $f | cut -c1-150; done
step 600 c3_bench.json python bench.py --workload stencil3d_32x32_c --steps 20 --warmup 10
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/c3_bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline_multiply"]["achieved"], d["roofline_multiply"]["frac"], d["roofline_multiply_native_api"]["achieved"], d["roofline_iteration"])
PY
