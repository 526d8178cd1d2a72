#!/bin/bash
# soak: repeated solves give identical bits -- small system (column operations folded into the producers' tails), P2 in mixed precision, P2 in z
source scripts/gpu_steps.sh
step 300 r03v_soak_small.txt python scripts/soak.py fd2d_16x16_z_small 600
step 300 r03v_soak_small_m.txt python scripts/soak.py fd2d_16x16_z_small 300 m
step 400 r03v_soak_m.txt python scripts/soak.py fd2d_16x16_z 60 m
step 400 r03v_soak_z.txt python scripts/soak.py fd2d_16x16_z 200
for f in small small_m m z; do tail -1 gpurun_out/r03v_soak_$f.txt; done
step 300 r03v_drv.log python -m pytest tests/test_bench_binary.py -q
tail -3 gpurun_out/r03v_drv.log
