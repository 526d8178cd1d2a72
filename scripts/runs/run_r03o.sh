#!/bin/bash
# do X-shaped vectors that lie exactly 2^k bytes apart hurt?  gap behind each vector (lab: TFQMRGPU_SKEW) on the power-of-two configurations
source scripts/gpu_steps.sh
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for wl in st:8:64:z:128:128:4 stencil2d_8x8_z st:16:16:z:128:128:32 fd2d_16x16_z; do
  for sk in 0 4352 69888 1118464; do
    echo "$wl TFQMRGPU_SKEW=$sk"; TFQMRGPU_SKEW=$sk timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu
  done
done
