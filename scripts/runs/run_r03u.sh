#!/bin/bash
# config 5 (8x8 z, 256 x 256 grid, 8 columns): caller-side tiled row numbering (library unchanged): does locality pay where the multiply may be fabric-bound?
source scripts/gpu_steps.sh
step 300 r03u_tiles4.txt python scripts/tiled_rows_probe.py 4 8
step 300 r03u_tiles16.txt python scripts/tiled_rows_probe.py 16 8
grep -v amdgpu gpurun_out/r03u_tiles4.txt; grep -v amdgpu gpurun_out/r03u_tiles16.txt
