#!/bin/bash
source scripts/gpu_steps.sh
for p in 1 0; do
step 300 c3_pre$p.txt env TFQMRGPU_EPI_PREFETCH=$p python scripts/bench_multiply.py stencil3d_32x32_c 20
step 300 c64_pre$p.txt env TFQMRGPU_EPI_PREFETCH=$p python scripts/bench_multiply.py st:64:64:c:24:24:4 20
step 300 c3264_pre$p.txt env TFQMRGPU_EPI_PREFETCH=$p python scripts/bench_multiply.py st:32:64:c:32:32:4 20
done
for f in gpurun_out/c3_pre*.txt gpurun_out/c64_pre*.txt gpurun_out/c3264_pre*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter|status" $f | cut -c1-150; done
