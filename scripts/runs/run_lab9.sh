#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
step 120 lab9_base_e1.txt $L scripts/lab/data/p2/base 1 v0,v6 20
step 120 lab9_base_e2.txt $L scripts/lab/data/p2/base 2 v0,v6 20
grep -h "^v" gpurun_out/lab9_*.txt
