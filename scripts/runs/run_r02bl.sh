#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do
step 600 ye_base_$r.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-multiply
step 600 ye_early_$r.json env TFQMRGPU_LIB=$PWD/scripts/bin/yearly/libtfQMRgpu.so python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-multiply
done
python3 - <<'PY'
import json
for t in ("base_1","early_1","base_2","early_2"):
    d=json.loads([l for l in open("gpurun_out/ye_%s.json" % t) if l.startswith("{")][-1])
    k=d["kernels"]
    print(t, d["value"], d["ms_per_step"], "v4", k["spmm_v4_dot"]["avg_ms"], "v5", k["spmm_v5_nrm_dot"]["avg_ms"], "it", d["roofline_iteration"]["ms_per_iteration"])
PY
