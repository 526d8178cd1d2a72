#!/bin/bash
# what the four-product default costs the z shapes above 16x16 (lab: TFQMRGPU_3M=0 | 1)
source scripts/gpu_steps.sh
L=tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for wl in st:32:32:z:64:64:8 st:64:64:z:32:32:8 st:16:32:z:90:90:8; do
  for m in 0 1; do echo "$wl TFQMRGPU_3M=$m"; TFQMRGPU_3M=$m timeout 300 python scripts/ab_fused.py $wl $L 2>&1 | grep -v amdgpu; done
done
