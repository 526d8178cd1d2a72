#!/bin/bash
# where do the small-block shapes stand? (product build; ~0.5 GB per vector in z, 0.25 GB in c)
source scripts/gpu_steps.sh
for wl in st:8:8:c:362:362:4 st:8:32:c:181:181:4 st:8:64:c:128:128:4 st:8:9:z:241:241:4 st:8:10:z:228:228:4 st:4:4:z:512:512:4 st:4:8:z:362:362:4 st:4:32:z:181:181:4 st:4:4:c:724:724:4 st:4:32:c:256:256:4; do
  echo "$wl"; timeout 300 python scripts/ab_fused.py $wl default 2>&1 | grep -v amdgpu
done
