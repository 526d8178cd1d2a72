#!/bin/bash
source scripts/gpu_steps.sh
step 600 tiled8.txt python scripts/tiled_rows_probe.py 4 8
grep -v amdgpu gpurun_out/tiled8.txt | tail -4
step 600 tiled16.txt python scripts/tiled_rows_probe.py 2 16
grep -v amdgpu gpurun_out/tiled16.txt | tail -4
