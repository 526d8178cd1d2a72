#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
step 120 lab13_onecol_e0.txt $L scripts/lab/data/onecol/base 0 v6,v6nt,v6nt2 20
step 120 lab13_onecol_e1.txt $L scripts/lab/data/onecol/base 1 v6,v6nt,v6nt2 20
grep -h "^v\|^#" gpurun_out/lab13_*.txt
