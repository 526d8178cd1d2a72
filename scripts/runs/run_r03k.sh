#!/bin/bash
# work-group descriptors for k_spmm_ilv16 (one scalar load instead of a chain of dependent ones): A/B in the lab build, parity
source scripts/gpu_steps.sh
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for d in 0 1 0 1; do
  echo "DESC=$d"; TFQMRGPU_DESC=$d timeout 300 python scripts/ab_fused.py fd2d_16x16_z $L 2>&1 | grep -v amdgpu
done
echo "small systems, DESC=0 then 1"
TFQMRGPU_LIB=$PWD/$L TFQMRGPU_DESC=0 timeout 300 python scripts/small_latency.py 2>&1 | grep -v amdgpu
TFQMRGPU_LIB=$PWD/$L TFQMRGPU_DESC=1 timeout 300 python scripts/small_latency.py 2>&1 | grep -v amdgpu
step 900 r03k_pytest.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_hash_mode.py tests/test_gpu_configs.py -q -x
tail -4 gpurun_out/r03k_pytest.log
