#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
step 120 lab10_base_e1.txt $L scripts/lab/data/p2/base 1 v0,v6 20
for g in c16 c16g16; do step 120 lab10_$g.txt $L scripts/lab/data/p2/$g 1 v6,v7 20; done
step 120 lab10_e2.txt $L scripts/lab/data/p2/c16 2 v6,v7 20
grep -h "^v" gpurun_out/lab10_*.txt
