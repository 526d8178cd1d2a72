#!/bin/bash
source scripts/gpu_steps.sh
for v in "X=0" "TFQMRGPU_CHUNK_KIB=8" "TFQMRGPU_CHUNK_KIB=32" "TFQMRGPU_ORDER_G=2" "TFQMRGPU_ORDER_G=8" "TFQMRGPU_ORDER_G=16" "TFQMRGPU_ORDER_BANDMULT=2" "TFQMRGPU_DEPTH=3"; do
  step 200 tune_${v//=/_}.txt env $v python scripts/bench_multiply.py fd2d_16x16_z 10
done
for f in gpurun_out/tune_*.txt; do echo "== $f"; grep -E "spmm|per iter|x_v6|xpay|v5_nrm" $f | cut -c1-110; done
