#!/bin/bash
source scripts/gpu_steps.sh
step 300 pin_base.txt python scripts/bench_multiply.py stencil3d_32x32_c 20
for v in pin0 pin8; do
step 300 $v.txt env TFQMRGPU_LIB=$PWD/scripts/bin/$v/libtfQMRgpu.so python scripts/bench_multiply.py stencil3d_32x32_c 20
done
step 300 pin_base_64.txt python scripts/bench_multiply.py st:64:64:c:24:24:4 20
step 300 pin0_64.txt env TFQMRGPU_LIB=$PWD/scripts/bin/pin0/libtfQMRgpu.so python scripts/bench_multiply.py st:64:64:c:24:24:4 20
for f in gpurun_out/pin*.txt; do echo "== $f"; grep -E "^multiply|spmm|per iter" $f | cut -c1-150; done
