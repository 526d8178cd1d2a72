#!/bin/bash
source scripts/gpu_steps.sh
L=scripts/bin/spmm_lab; D=scripts/lab/data/p2
step 120 lab2_c16_e1.txt $L $D/c16 1 v0,v2,v2a1,v2a2,v2w8,v2w8a1,v2w8a2 20
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
step 200 lab2_pmc_fetch.txt rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/lab2_pmc1 -- $L $D/c16 1 v0,v2 2
step 200 lab2_pmc_write.txt rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/lab2_pmc2 -- $L $D/c16 1 v0,v2 2
step 200 lab2_pmc_fetch_base.txt rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/lab2_pmc3 -- $L $D/base 1 v0 2
cat gpurun_out/lab2_c16_e1.txt
find gpurun_out/lab2_pmc1 gpurun_out/lab2_pmc2 gpurun_out/lab2_pmc3 -name "*counter_collection.csv" | head
