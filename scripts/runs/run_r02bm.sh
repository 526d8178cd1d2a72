#!/bin/bash
source scripts/gpu_steps.sh
step 1100 pytest_bm.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_bm.log | tail -6
for wl in fd2d_16x16_z stencil3d_32x32_c stencil2d_8x8_z; do
step 600 v5_$wl.json python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-multiply
done
python3 - <<'PY'
import json
for wl in ("fd2d_16x16_z", "stencil3d_32x32_c", "stencil2d_8x8_z"):
    d=json.loads([l for l in open("gpurun_out/v5_%s.json" % wl) if l.startswith("{")][-1])
    print(wl, d["value"], d["ms_per_step"], d["iterations_per_solve"], d["roofline"]["frac"], d["roofline"]["avg_ms"], d["roofline_iteration"]["ms_per_iteration"])
PY
