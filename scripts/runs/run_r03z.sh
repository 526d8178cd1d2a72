#!/bin/bash
# occupancy probe of the fused 16x16 z multiplies: (a) two waves per SIMD through an unused 60 KiB of dynamic LDS per work group (lab switch),
# (b) epilogue operands through LDS-DMA (EPI 2: 128 VGPRs = four waves per SIMD, 24 bytes of scratch; EPI 1 stays at 148 = three waves)
source scripts/gpu_steps.sh
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for k in 0 60 0 60; do echo "TFQMRGPU_ILV16_LDS_KIB=$k"; TFQMRGPU_ILV16_LDS_KIB=$k timeout 300 python scripts/ab_fused.py fd2d_16x16_z $L 2>&1 | grep -v amdgpu; done
timeout 600 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_elds.so tfqmrgpu_amd/lib/libtfQMRgpu.so scripts/bin/libtfQMRgpu_elds.so 2>&1 | grep -v amdgpu
