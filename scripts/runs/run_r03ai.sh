#!/bin/bash
# fold crossover, small end: 16 ... 256 chunks
source scripts/gpu_steps.sh
timeout 800 python scripts/fold_crossover.py st:16:16:z:8:8:1 st:16:16:z:8:8:2 st:16:16:z:12:12:2 st:16:16:z:16:16:2 st:16:16:z:16:16:4 st:8:8:z:8:8:4 st:8:8:z:16:16:4 st:8:8:z:24:24:4 st:32:32:c:8:8:2 st:4:4:z:32:32:4 FD:1.75,6.75,2,3,0.0,4 2>&1 | grep -v amdgpu
