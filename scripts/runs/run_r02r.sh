#!/bin/bash
source scripts/gpu_steps.sh
step 1100 pytest_r02r.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_r02r.log | tail -40
