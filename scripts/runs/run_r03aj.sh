#!/bin/bash
# fold crossover on a created stream (scripts/small_latency.py's case) and on the null stream
source scripts/gpu_steps.sh
for st in created null; do
FOLD_STREAM=$st timeout 500 python scripts/fold_crossover.py st:16:16:z:8:8:1 st:16:16:z:8:8:2 st:16:16:z:16:16:2 st:16:16:z:16:16:4 st:8:8:z:16:16:4 st:8:8:z:24:24:4 FD:1.75,6.75,2,3,0.0,4 FD:6,24,4,2,-0.25,4 st:16:16:z:32:32:4 2>&1 | grep -v amdgpu
done
