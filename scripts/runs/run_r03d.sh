#!/bin/bash
# round 3: mixed precision 'm', frozen switches (lab build), 3M opt-in: the whole GPU suite, mixed tests first
source scripts/gpu_steps.sh
step 900 r03d_mixed.log python -m pytest tests/test_gpu_mixed.py -q -x
tail -30 gpurun_out/r03d_mixed.log
step 1100 r03d_pytest.log python -m pytest tests -m gpu -q --deselect tests/test_gpu_mixed.py
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r03d_pytest.log | tail -15
