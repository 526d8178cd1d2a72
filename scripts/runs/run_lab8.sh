#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
for g in c12 c12g16; do step 120 lab8_$g.txt $L scripts/lab/data/p2_12/$g 1 v0,v5c12,v5c12u3 20; done
for g in c8 c8g16; do step 120 lab8_$g.txt $L scripts/lab/data/p2_8/$g 1 v0,v5c8,v5c8u2 20; done
step 120 lab8_c16.txt $L scripts/lab/data/p2/c16 1 v5u1 20
step 120 lab8_e2.txt $L scripts/lab/data/p2_8/c8 2 v0,v5c8,v5c8u2 20
grep -h "^v" gpurun_out/lab8_*.txt
