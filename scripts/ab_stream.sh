#!/bin/bash
# A/B of library builds / tuning switches on the multiply: scripts/ab_stream.sh  (WORKLOADS, VARIANTS, LIBS from the environment)
mkdir -p gpurun_out
NEW=$PWD/tfqmrgpu_amd/lib/libtfQMRgpu.so; OLD=${OLD:-$NEW}   # OLD: a build of another commit, e.g. from a git worktree
run() { echo "== $1 | $(basename $2) | $3"; env $3 TFQMRGPU_LIB=$2 python scripts/bench_multiply.py $1 10 2>&1 | grep -E "^multiply|spmm|per iter|solve status" | sed -e 's/"peak[^}]*//'; }
for wl in ${WORKLOADS:-fd2d_16x16_z}; do
  run $wl $OLD "X=0"
  for lib in $NEW $LIBS; do
    for v in ${VARIANTS:-X=0}; do run $wl $lib "$v"; done
  done
done
