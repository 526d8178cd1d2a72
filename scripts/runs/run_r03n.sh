#!/bin/bash
# 8 x 32 and 8 x 64 complex<double> on the row-pair-interleaved order (k_spmm_ilv8w): parity, then A/B against the native order (lab: TFQMRGPU_ILV=16)
source scripts/gpu_steps.sh
step 900 r03n_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_hash_mode.py -q -x
tail -4 gpurun_out/r03n_pytest.log
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for wl in st:8:32:z:181:181:4 st:8:64:z:128:128:4; do
  for ilv in 16 1 16 1; do
    echo "$wl TFQMRGPU_ILV=$ilv"; TFQMRGPU_ILV=$ilv timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
  done
done
