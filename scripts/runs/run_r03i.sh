#!/bin/bash
# chunk length and the fused multiplies (lab build): does a wave with several Y blocks per work group run the multiplies faster?
source scripts/gpu_steps.sh
for kib in 16 32 64 128; do
  echo "CHUNK_KIB=$kib"
  TFQMRGPU_CHUNK_KIB=$kib timeout 300 python scripts/ab_fused.py fd2d_16x16_z tfqmrgpu_amd/lib/libtfQMRgpu_lab.so 2>&1 | grep -v amdgpu
done
