#!/usr/bin/env python3
"""Timeline of one small solve from a rocprofv3 --kernel-trace (+ --memory-copy-trace) run: per kernel duration and the gap to the previous
activity on the stream.  usage: python3 scripts/trace_gaps.py <dir> [first_n]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void tfq::", "")[:60]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the last solve: find the last k_init_col
starts = [i for i, e in enumerate(ev) if "k_init_col" in e[2]]
i0 = starts[-1] if starts else 0
prev = ev[i0][0]
for s, e, k in ev[i0:i0 + n]:
    print("%9.2f us  gap %7.2f  dur %7.2f  %s" % ((s - ev[i0][0]) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, k))
    prev = e
