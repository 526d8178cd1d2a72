#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab
for g in base base_xhot base_ahot base_axhot; do step 120 lab14_$g.txt $L scripts/lab/data/p2/$g 1 v6 20; done
grep -h "^v\|^#" gpurun_out/lab14_*.txt
