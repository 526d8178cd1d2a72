#!/bin/bash
source scripts/gpu_steps.sh
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for wl in st:8:64:z:128:128:4 st:8:64:z:127:127:4; do
  for ilv in 16 1 16 1; do
    echo "$wl TFQMRGPU_ILV=$ilv"; TFQMRGPU_ILV=$ilv timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
  done
done
echo "fd2d_16x16_z: lab against product"
timeout 300 python scripts/ab_fused.py fd2d_16x16_z $L default 2>&1 | grep -v amdgpu
