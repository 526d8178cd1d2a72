#!/bin/bash
# 8 x 8 | 32 | 64 complex<float> on the quad-interleaved order (k_spmm_ilv8f): parity, then A/B against the native order (lab: TFQMRGPU_ILV=3)
source scripts/gpu_steps.sh
step 900 r03r_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_hash_mode.py -q -x
tail -12 gpurun_out/r03r_pytest.log
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for wl in st:8:8:c:362:362:4 st:8:32:c:181:181:4 st:8:64:c:128:128:4; do
  for ilv in 3 1 3 1; do
    echo "$wl TFQMRGPU_ILV=$ilv"; TFQMRGPU_ILV=$ilv timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
  done
done
