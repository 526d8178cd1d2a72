#!/bin/bash
# round 3: PMC of the fused multiplies under different row orders / column-group sizes; the new P2 test; bench with the two byte models
source scripts/gpu_steps.sh
step 400 r03b_p2test.log python -m pytest tests/test_gpu_configs.py -q -k "config2" -x
step 400 r03b_bench.json python bench.py --steps 5 --warmup 2 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/r03b_pmc
step 600 r03b_pmc.log rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d gpurun_out/r03b_pmc -- python3 scripts/row_order_probe.py 4,8,16,49 natural,raster,strip8,tile4 0
python3 scripts/pmc_by_kernel.py gpurun_out/r03b_pmc k_spmm_ilv16 14 > gpurun_out/r03b_pmc_summary.txt 2>&1
rm -rf gpurun_out/r03b_pmc
cat gpurun_out/r03b_pmc_summary.txt | cut -c1-200
