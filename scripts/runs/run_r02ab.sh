#!/bin/bash
source scripts/gpu_steps.sh
for r in 1 2; do
step 300 hashc_new_$r.txt python scripts/bench_multiply.py st:16:16:c:96:96:16 10
step 300 hashc_inloop_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/abl/libtfQMRgpu.so python scripts/bench_multiply.py st:16:16:c:96:96:16 10
step 300 hashc_old_$r.txt env TFQMRGPU_LIB=$PWD/scripts/bin/old/libtfQMRgpu.so python scripts/bench_multiply.py st:16:16:c:96:96:16 10
done
step 300 hashab_final.txt python scripts/bench_multiply.py fd2d_16x16_z 5
for f in gpurun_out/hashc_*.txt gpurun_out/hashab_final.txt; do echo "== $f"; grep -E "spmm|per iter" $f | cut -c1-70; done
