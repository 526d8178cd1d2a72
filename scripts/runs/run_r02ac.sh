#!/bin/bash
source scripts/gpu_steps.sh
run() { tag=$1; shift; step 300 c5_$tag.txt env "$@" python scripts/bench_multiply.py stencil2d_8x8_z 5; }
run base X=0
run g8 TFQMRGPU_ORDER_G=8
run g2 TFQMRGPU_ORDER_G=2
run ch256 TFQMRGPU_CHUNK_KIB=256
run ch128 TFQMRGPU_CHUNK_KIB=128
run ch32 TFQMRGPU_CHUNK_KIB=32
run ch256g8 TFQMRGPU_CHUNK_KIB=256 TFQMRGPU_ORDER_G=8
run ch32g8 TFQMRGPU_CHUNK_KIB=32 TFQMRGPU_ORDER_G=8
for f in gpurun_out/c5_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-90; done
