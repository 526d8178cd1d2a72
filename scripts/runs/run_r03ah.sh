#!/bin/bash
# where does the fold of the column operations stop paying (Plan::foldOk = chunks <= 512 was set between two data points)
source scripts/gpu_steps.sh
timeout 800 python scripts/fold_crossover.py 2>&1 | grep -v amdgpu
