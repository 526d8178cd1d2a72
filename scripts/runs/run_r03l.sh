#!/bin/bash
source scripts/gpu_steps.sh
step 300 r03l_clock.txt python scripts/clock_under_load.py
grep -v amdgpu gpurun_out/r03l_clock.txt
