#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_32f.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_configs.py tests/test_gpu_operator.py -m gpu -q -x
grep -E "^FAILED|passed|failed|Error" gpurun_out/pytest_32f.log | tail -6
step 300 c3_ilv1.txt python scripts/bench_multiply.py stencil3d_32x32_c 20
step 300 c3_ilv3.txt env TFQMRGPU_ILV=3 python scripts/bench_multiply.py stencil3d_32x32_c 20
for f in gpurun_out/c3_ilv*.txt; do echo "== $f"; grep -E "^multiply|spmm|xpay|v5_nrm|x_v6|per iter|status" $f | cut -c1-150; done
