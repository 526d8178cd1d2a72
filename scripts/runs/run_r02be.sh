#!/bin/bash
source scripts/gpu_steps.sh
step 600 cmp_first.txt python scripts/bin/cmp_first.py
cat gpurun_out/cmp_first.txt | grep -v amdgpu
