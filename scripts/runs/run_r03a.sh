#!/bin/bash
# round 3, first call: baseline bench on this round's box + the row-order probe
source scripts/gpu_steps.sh
step 500 r03a_bench.json python bench.py --steps 5 --warmup 2
step 900 r03a_row_order.txt python scripts/row_order_probe.py 4,8,12,16,24
