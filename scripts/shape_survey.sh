#!/bin/bash
# all 15 block shapes of allowed_block_sizes.h in both precisions on 5-point stencil systems of about 256 MB per vector:
# multiply rate and iteration roofline fraction (to spot shapes whose kernels fall behind)
mkdir -p gpurun_out
for prec in z c; do
for s in "4 4" "4 5" "4 8" "4 32" "8 8" "8 9" "8 10" "8 32" "8 64" "16 16" "16 32" "16 64" "32 32" "32 64" "64 64"; do
  set -- $s; lm=$1; ln=$2
  bytes=$((2*lm*ln*( $( [ $prec = z ] && echo 8 || echo 4 ) )))
  # blocks for ~256 MB: nx*ny*ncols; 4 columns
  nb=$((268435456/bytes)); n=$(python3 -c "import math;print(max(16,int(math.sqrt($nb/4))))")
  wl="st:$lm:$ln:$prec:$n:$n:4"
  python scripts/bench_multiply.py $wl 5 2>&1 | grep -E "^multiply|per iteration|spmm_v4|x_v6_v7|solve status" | sed -e 's/"peak[^}]*//' | tr '\n' ' ' | sed -e "s/^/$wl /"; echo
done; done
