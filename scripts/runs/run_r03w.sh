#!/bin/bash
source scripts/gpu_steps.sh
for wl in st:4:4:z:512:512:4 st:4:4:c:724:724:4; do echo "$wl"; timeout 300 python scripts/ab_fused.py $wl default 2>&1 | grep -v amdgpu; done
step 900 r03w_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_operator.py -q -x
tail -3 gpurun_out/r03w_pytest.log
