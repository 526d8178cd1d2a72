#!/bin/bash
source scripts/gpu_steps.sh
step 300 hot_c3.txt python scripts/hot_operand_probe.py stencil3d_32x32_c 20
step 300 hot_64c.txt python scripts/hot_operand_probe.py st:64:64:c:24:24:4 20
step 300 hot_p2.txt python scripts/hot_operand_probe.py fd2d_16x16_z 20
step 300 hot_64z.txt python scripts/hot_operand_probe.py st:64:64:z:24:24:4 20
cat gpurun_out/hot_*.txt | grep -v amdgpu.ids
