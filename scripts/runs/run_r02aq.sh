#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_aq.log python -m pytest tests/test_gpu_operator.py tests/test_gpu_parity.py -m gpu -q -x -k "operator_that_repeats or set_get_matrix_layouts"
grep -E "^FAILED|passed|failed|Error" gpurun_out/pytest_aq.log | tail -5
