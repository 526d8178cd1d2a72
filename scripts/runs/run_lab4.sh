#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab; D=scripts/lab/data/p2
for g in c16g2 c16 c16g8 c16g16 c16g49; do
step 120 lab4_$g.txt $L $D/$g 1 v3 20
done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 200 lab4_pmc_g16.txt rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/lab4_pmc_g16 -- $L $D/c16g16 1 v3 2
step 200 lab4_pmc_g49.txt rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/lab4_pmc_g49 -- $L $D/c16g49 1 v3 2
grep -h "^v3" gpurun_out/lab4_c16*.txt
