#!/bin/bash
source scripts/gpu_steps.sh
step 900 pytest_r02i.log python -m pytest tests -m gpu -q
grep -E "passed|failed|FAILED" gpurun_out/pytest_r02i.log | tail -8
