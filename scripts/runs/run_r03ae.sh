#!/bin/bash
# native-API multiply (plain mode of k_spmm_mfma): contiguous eighths of the caller's Y blocks per XCD (lab switch TFQMRGPU_PLAIN_XCD) on config 1 and on P2
source scripts/gpu_steps.sh
export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for x in 0 1 0 1; do
  echo "TFQMRGPU_PLAIN_XCD=$x"
  for prec in f z; do TFQMRGPU_PLAIN_XCD=$x python -m tfqmrgpu_amd.bench_tfqmrgpu multi tests/golden/plan_unordered.14-287-16.gz $prec 20 5 2>&1 | grep "GPU performance\|maxdev"; done
done
for x in 0 1; do echo "P2 TFQMRGPU_PLAIN_XCD=$x"; TFQMRGPU_PLAIN_XCD=$x timeout 300 python scripts/bench_multiply.py fd2d_16x16_z 5 2>&1 | grep -E "^multiply"; done
