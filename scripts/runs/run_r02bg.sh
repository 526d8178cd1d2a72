#!/bin/bash
source scripts/gpu_steps.sh
step 1100 pytest_bg.log python -m pytest tests -m gpu -q
grep -E "^FAILED|passed|failed" gpurun_out/pytest_bg.log | tail -8
step 300 soak_c5b.txt python scripts/soak.py stencil2d_8x8_z 50
tail -1 gpurun_out/soak_c5b.txt
step 300 soak_p2b.txt python scripts/soak.py fd2d_16x16_z 50
tail -1 gpurun_out/soak_p2b.txt
