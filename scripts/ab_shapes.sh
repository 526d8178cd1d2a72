#!/bin/bash
# A/B of library builds over block shapes: scripts/ab_shapes.sh "workload ..." libA.so libB.so
wls=$1; shift
for wl in $wls; do
  for lib in "$@"; do
    echo "== $wl $(basename $lib)"
    TFQMRGPU_LIB=$lib python scripts/bench_multiply.py $wl 10 2>&1 | grep -E "^multiply|spmm|per iter|solve status" | sed -e 's/"peak[^}]*//'
  done
done
