#!/bin/bash
source scripts/gpu_steps.sh
step 300 pytest_getter.log python -m pytest tests/test_gpu_hash_mode.py -m gpu -q -x -k "getter"
tail -2 gpurun_out/pytest_getter.log
rm -rf gpurun_out/r02_pmc gpurun_out/r02_pmc8
step 900 pmc_p2.log bash scripts/pmc_collect.sh gpurun_out/r02_pmc fd2d_16x16_z
step 900 pmc_8x8.log bash scripts/pmc_collect.sh gpurun_out/r02_pmc8 stencil2d_8x8_z
python3 scripts/pmc_summary.py gpurun_out/r02_pmc > gpurun_out/r02_pmc_summary.json
python3 scripts/pmc_summary.py gpurun_out/r02_pmc8 > gpurun_out/r02_pmc_summary_8x8z.json
python3 - <<'PY'
import json
for f in ("gpurun_out/r02_pmc_summary.json", "gpurun_out/r02_pmc_summary_8x8z.json"):
    d = json.load(open(f))
    for k, v in d.items():
        if "spmm" in k or "x_v6" in k or "xpay" in k or "v5_nrm" in k:
            print(f[-14:], k[:60], v.get("avg_us_working"), v.get("hbm_read_MB(2x FETCH_SIZE)"), v.get("hbm_write_MB"), v.get("l2_hit_rate"))
PY
