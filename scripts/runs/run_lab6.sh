#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
for g in c16 c16g16 c16g49; do step 120 lab6_$g.txt $L scripts/lab/data/p2/$g 1 v3,v4 20; done
step 120 lab6_e2.txt $L scripts/lab/data/p2/c16g16 2 v3,v4 20
grep -h "^v" gpurun_out/lab6_*.txt
