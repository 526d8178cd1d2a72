#!/bin/bash
# where requests queue: average L1 -> L2 read latency (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ) and average L2 -> fabric read latency
# (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, Little's law) per kernel of a P2 solve; separate rocprofv3 --pmc passes.  usage: scripts/pmc_latency.sh <outdir> [workload]
out=${1:-gpurun_out/pmc_lat}; wl=${2:-fd2d_16x16_z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
rocprofv3 --list-avail > "$out/avail.txt" 2>&1
grep -o "TCP_TCC_READ_REQ_LATENCY[_a-z]*\|TCP_TCC_READ_REQ_sum\|TCC_EA0_RDREQ_LEVEL[_a-z]*\|TCP_PENDING_STALL_CYCLES[_a-z]*\|TCC_TAG_STALL[_a-z]*\|TCC_BUSY[_a-z]*\|TCP_TCC_WRITE_REQ_LATENCY[_a-z]*\|TCC_EA0_WRREQ_LEVEL[_a-z]*\|TCC_EA0_WRREQ_sum\|TCP_TA_TCP_STATE_READ[_a-z]*\|TCP_READ_TAGCONFLICT_STALL_CYCLES[_a-z]*\|TA_BUSY[_a-z]*\|TCP_TCC_WRITE_REQ_sum" "$out/avail.txt" | sort -u | tr '\n' ' '; echo
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCC_TAG_STALL_sum TCC_BUSY_sum" "TA_BUSY_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$out/pass$i" -- python3 scripts/pmc_driver.py "$wl" 2 > "$out/pass$i.log" 2>&1 || { echo "pass $i ($set) failed"; tail -3 "$out/pass$i.log"; continue; }
  python3 scripts/pmc_by_kernel.py "$out/pass$i" k_ | grep -E "^#|k_spmm_ilv16<1, true, false, false>|k_spmm_ilv16<2|k_x_v6_v7|k_xpay_v6|k_v5_nrm" | awk 'NR==1 || $1==1' | cut -c1-220
done
