#!/bin/bash
# round 3: fresh PMC passes of the bench workload (profiles/r03_pmc_summary.json, pmc_traffic.json), mixed-precision ranks test
source scripts/gpu_steps.sh
step 600 r03m_ranks.log python -m pytest tests/test_gpu_ranks.py tests/test_gpu_mixed.py -q
tail -3 gpurun_out/r03m_ranks.log
rm -rf gpurun_out/pmc
bash scripts/pmc_collect.sh gpurun_out/pmc fd2d_16x16_z
python3 scripts/pmc_summary.py gpurun_out/pmc > gpurun_out/r03m_pmc_summary.json
find gpurun_out/pmc -name "*.csv" -delete
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03m_pmc_summary.json"))
for k, v in d.items():
    if "spmm" in k or "x_v6" in k or "xpay" in k or "v5_nrm" in k:
        print(k[:60], v.get("avg_us_working"), v.get("hbm_read_MB(2x FETCH_SIZE)"), v.get("hbm_write_MB"), v.get("l2_hit_rate"), v.get("mfma_busy_per_sq_busy"))
PY
