#!/bin/bash
source scripts/gpu_steps.sh
step 600 soak_p2.txt python scripts/soak.py fd2d_16x16_z 100
tail -1 gpurun_out/soak_p2.txt
step 600 soak_c3.txt python scripts/soak.py stencil3d_32x32_c 300
tail -1 gpurun_out/soak_c3.txt
step 600 soak_c5.txt python scripts/soak.py stencil2d_8x8_z 100
tail -1 gpurun_out/soak_c5.txt
step 600 soak_c16.txt python scripts/soak.py st:16:16:c:96:96:16 200
tail -1 gpurun_out/soak_c16.txt
