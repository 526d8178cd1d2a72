#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
step 120 lab12_onecol_e0.txt $L scripts/lab/data/onecol/base 0 v0,v6,v8 20
step 120 lab12_p2_e0.txt $L scripts/lab/data/p2/base 0 v0,v6,v8 20
grep -h "^v\|^#" gpurun_out/lab12_*.txt
