#!/bin/bash
source scripts/gpu_steps.sh
step 1100 pytest_ax.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed|Error" gpurun_out/pytest_ax.log | tail -6
