#!/bin/bash
source scripts/gpu_steps.sh
run() { tag=$1; shift; step 300 trav_$tag.txt env "$@" python scripts/bench_multiply.py fd2d_16x16_z 5; }
run base X=0
run k3o TFQMRGPU_TRAV_K3=1
run k3or TFQMRGPU_TRAV_K3=3
run k3r TFQMRGPU_TRAV_K3=2
run k4o TFQMRGPU_TRAV_K4=1
run k4or TFQMRGPU_TRAV_K4=3
run k1o TFQMRGPU_TRAV_K1=1
run k1or TFQMRGPU_TRAV_K1=3
run allo TFQMRGPU_TRAV_K1=1 TFQMRGPU_TRAV_K3=1 TFQMRGPU_TRAV_K4=1
for f in gpurun_out/trav_*.txt; do echo "== $f"; grep -E "xpay|v5_nrm|x_v6|spmm|per iter|status" $f | cut -c1-62; done
