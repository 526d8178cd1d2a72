#!/bin/bash
source scripts/gpu_steps.sh
step 300 hash_base.txt python scripts/bench_multiply.py fd2d_16x16_z 10
step 300 hash_abl.txt env TFQMRGPU_LIB=$PWD/scripts/bin/abl/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 10
step 300 hash_base_c.txt python scripts/bench_multiply.py st:16:16:c:96:96:16 10
step 300 hash_abl_c.txt env TFQMRGPU_LIB=$PWD/scripts/bin/abl/libtfQMRgpu.so python scripts/bench_multiply.py st:16:16:c:96:96:16 10
step 300 hash_base_8.txt python scripts/bench_multiply.py stencil2d_8x8_z 10
step 300 hash_abl_8.txt env TFQMRGPU_LIB=$PWD/scripts/bin/abl/libtfQMRgpu.so python scripts/bench_multiply.py stencil2d_8x8_z 10
for f in gpurun_out/hash_*.txt; do echo "== $f"; grep -E "spmm|per iter|solve status" $f | cut -c1-150; done
