#!/bin/bash
# column-group size of the launch order where A is large against one X column (config 5: 8x8 z, 8 columns; config 3: 2 columns)
source scripts/gpu_steps.sh
for g in 4 8 2; do
  echo "stencil2d_8x8_z ORDER_G=$g"
  TFQMRGPU_ORDER_G=$g timeout 300 python scripts/ab_fused.py stencil2d_8x8_z tfqmrgpu_amd/lib/libtfQMRgpu_lab.so 2>&1 | grep -v amdgpu
done
for g in 4 8 16; do
  echo "st:16:16:z:128:128:32 (config 4 shard) ORDER_G=$g"
  TFQMRGPU_ORDER_G=$g timeout 300 python scripts/ab_fused.py st:16:16:z:128:128:32 tfqmrgpu_amd/lib/libtfQMRgpu_lab.so 2>&1 | grep -v amdgpu
done
