#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab; D=scripts/lab/data/p2
step 120 lab3_c16_e1.txt $L $D/c16 1 v0,v2,v3 20
step 120 lab3_c16_e2.txt $L $D/c16 2 v0,v2,v3 20
step 120 lab3_line16_e1.txt $L $D/line16 1 v0,v3 20
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 200 lab3_pmc_fetch.txt rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/lab3_pmc1 -- $L $D/c16 1 v3 2
cat gpurun_out/lab3_c16_e1.txt gpurun_out/lab3_c16_e2.txt gpurun_out/lab3_line16_e1.txt
