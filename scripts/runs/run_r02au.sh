#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do
step 300 ded_gen_$r.txt python scripts/bench_multiply.py stencil3d_32x32_c 10
step 300 ded_ded_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/ded/libtfQMRgpu.so python scripts/bench_multiply.py stencil3d_32x32_c 10
done
for f in gpurun_out/ded_*.txt; do echo "== $f"; grep -E "spmm|per iter" $f | cut -c1-100; done
