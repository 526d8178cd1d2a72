#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc passes (scripts/pmc_collect.sh): per kernel the average counter values per launch
that did work, with the HBM byte estimate of /opt/skills/guides/MI355X_MICROARCH.md (section HBM):
FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950 for wide streaming reads -> doubled;
WRITE_SIZE (KiB) is exact for 16-B-per-lane streaming stores.
usage: python3 scripts/pmc_summary.py <dir>"""
import csv, glob, os, sys, collections, json
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "pass*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(d, "pass1", "*", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {}
for k in sorted(agg):
    if not k.startswith("void tfq::k_") or "convert" in k: continue
    us = dur.get(k, [])
    work = [u for u in us if u > 20.0]      # gated-off launches finish in a few microseconds
    row = {"launches": len(us), "working_launches": len(work), "avg_us_working": round(sum(work) / max(1, len(work)), 1)}
    for c, vals in agg[k].items():
        big = sorted(vals)[len(vals) // 2:]  # the working launches carry the large values
        row[c] = round(sum(big) / len(big), 1)
    if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
        row["hbm_read_MB(2x FETCH_SIZE)"] = round(2 * row["FETCH_SIZE"] * 1024 / 1e6, 1)
        row["hbm_write_MB"] = round(row["WRITE_SIZE"] * 1024 / 1e6, 1)
    if "TCC_HIT_sum" in row: row["l2_hit_rate"] = round(row["TCC_HIT_sum"] / max(1.0, row["TCC_HIT_sum"] + row["TCC_MISS_sum"]), 3)
    if "TCC_EA0_RDREQ_sum" in row: row["dram_share_of_fabric_reads"] = round(row["TCC_EA0_RDREQ_DRAM_sum"] / max(1.0, row["TCC_EA0_RDREQ_sum"]), 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "SQ_BUSY_CYCLES" in row: row["mfma_busy_per_sq_busy"] = round(row["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1.0, row["SQ_BUSY_CYCLES"]), 3)
    out[k.replace("void tfq::", "")] = row
print(json.dumps(out, indent=1))
