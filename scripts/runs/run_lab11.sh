#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
for g in base baseg8 baseg16 base8; do step 120 lab11_$g.txt $L scripts/lab/data/p2/$g 1 v0,v6,v8,v8w4 20; done
step 120 lab11_e2.txt $L scripts/lab/data/p2/base 2 v0,v6,v8 20
grep -h "^v" gpurun_out/lab11_*.txt
