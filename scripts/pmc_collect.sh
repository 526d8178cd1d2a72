#!/bin/bash
# collects PMC counters of the bench workload in separate passes (rocprofv3 allows few TCC/SQ counters per pass)
# usage: scripts/pmc_collect.sh <outdir> [workload]
out=${1:-gpurun_out/pmc}; wl=${2:-fd2d_16x16_z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 scripts/pmc_driver.py "$wl" 3 > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; exit 1; }
done
find "$out" -name "*counter_collection.csv" | head
