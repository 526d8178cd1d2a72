#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02o.log python -m pytest tests -m gpu -q -x
for i in 1 2; do step 300 c16_ilv$i.txt env TFQMRGPU_ILV=$i python scripts/bench_multiply.py st:16:16:c:96:96:16 10; done
grep -E "passed|failed|FAILED" gpurun_out/pytest_r02o.log | tail -5
for f in gpurun_out/c16_ilv*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter|solve status|xpay|x_v6|v5_nrm" $f | cut -c1-150; done
