#!/bin/bash
source scripts/gpu_steps.sh
step 600 c5.log python bench.py --workload stencil2d_8x8_z --steps 3 --warmup 1
tail -c 1500 gpurun_out/c5.log
