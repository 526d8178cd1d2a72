#!/bin/bash
# native-API multiply: round-robin work groups against contiguous eighths per XCD (TFQMRGPU_PLAIN_XCD) over sizes and shapes
source scripts/gpu_steps.sh
export TFQMRGPU_LIB=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu_lab.so
for wl in st:16:16:z:24:24:4 st:16:16:z:48:48:4 st:16:16:z:128:128:4 st:16:16:z:64:64:32 st:16:16:c:48:48:8 st:16:16:c:181:181:4 st:32:32:z:64:64:4 st:32:32:c:32:32:4 st:64:64:z:16:16:4 fd2d_16x16_z_small; do
  for x in 0 1; do echo -n "$wl PLAIN_XCD=$x "; TFQMRGPU_PLAIN_XCD=$x timeout 300 python scripts/bench_multiply.py $wl 5 2>&1 | grep -E "^multiply" | cut -c1-200; done
done
