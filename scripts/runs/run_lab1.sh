#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab; D=scripts/lab/data/p2
step 120 lab1_base_e1.txt $L $D/base 1 v0 20
step 120 lab1_base_e2.txt $L $D/base 2 v0 20
step 120 lab1_line16_e1.txt $L $D/line16 1 v0,v2,v2b 20
step 120 lab1_c16_e1.txt $L $D/c16 1 v0,v2,v2b 20
step 120 lab1_c16_e2.txt $L $D/c16 2 v0,v2,v2b 20
cat gpurun_out/lab1_*.txt
