#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
for g in c12 c12g16 c12g49; do step 120 lab5_$g.txt $L scripts/lab/data/p2_12/$g 1 v0,v3c12 20; done
for g in c8 c8g16 c8g49; do step 120 lab5_$g.txt $L scripts/lab/data/p2_8/$g 1 v0,v3c8 20; done
grep -h "^v" gpurun_out/lab5_*.txt
