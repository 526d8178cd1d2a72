#!/bin/bash
source scripts/gpu_steps.sh
step 300 prof_overhead.txt python scripts/prof_overhead.py fd2d_16x16_z 5
cat gpurun_out/prof_overhead.txt | tail -5
step 900 pytest_r02t.log python -m pytest tests -m gpu -q -x
grep -E "^FAILED|passed|failed" gpurun_out/pytest_r02t.log | tail -5
