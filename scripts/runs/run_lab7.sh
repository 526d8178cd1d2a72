#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
for g in c16 c16g16 c16g49; do step 120 lab7_$g.txt $L scripts/lab/data/p2/$g 1 v3,v5 20; done
step 120 lab7_e2.txt $L scripts/lab/data/p2/c16g16 2 v3,v5 20
grep -h "^v" gpurun_out/lab7_*.txt
