#!/bin/bash
source scripts/gpu_steps.sh
step 300 nold_base.txt python scripts/bench_multiply.py stencil3d_32x32_c 20
step 300 nold_abl.txt env TFQMRGPU_LIB=$PWD/scripts/bin/nold/libtfQMRgpu.so python scripts/bench_multiply.py stencil3d_32x32_c 20
for f in gpurun_out/nold_*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-150; done
