#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02g.log python -m pytest tests -m gpu -q -x
for i in 1 0; do step 300 b8_ilv$i.txt env TFQMRGPU_ILV=$i python scripts/bench_multiply.py stencil2d_8x8_z 10; done
step 400 bench_r02g.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline
grep -E "passed|failed|FAILED" gpurun_out/pytest_r02g.log | tail -8
for f in gpurun_out/b8_ilv*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter|solve status" $f | cut -c1-170; done
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_r02g.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"])
for k in ("roofline","roofline_multiply","roofline_multiply_native_api","roofline_multiply_hbm_bound","roofline_iteration"):
    r=d[k]; print(k, r.get("avg_ms", r.get("ms_per_iteration")), r["achieved"], r["unit"], r["frac"])
PY
