#!/bin/bash
source scripts/gpu_steps.sh
step 300 nored_base.txt python scripts/bench_multiply.py fd2d_16x16_z 5
step 300 nored_abl.txt env TFQMRGPU_LIB=$PWD/scripts/bin/abl/libtfQMRgpu.so python scripts/bench_multiply.py fd2d_16x16_z 5
for f in gpurun_out/nored_*.txt; do echo "== $f"; grep -E "xpay|v5_nrm|x_v6|per iter|status" $f | cut -c1-100; done
