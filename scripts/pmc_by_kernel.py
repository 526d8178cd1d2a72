#!/usr/bin/env python3
"""Counters of one rocprofv3 --pmc pass per SOLVE (a solve starts with k_init_col) and kernel: mean over the launches that did work.
usage: python3 scripts/pmc_by_kernel.py <dir> <kernel substring>"""
import csv, glob, os, sys, collections
d, sub = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"])
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({c for v in rows.values() for c in v})
print("# solve kernel working_launches " + " ".join(names))
solve = -1
groups = collections.OrderedDict()
for (disp, kn) in sorted(rows):
    if "k_init_col" in kn: solve += 1
    if sub in kn: groups.setdefault((solve, kn), []).append(rows[(disp, kn)])
for (sv, kn), grp in groups.items():
    ref = max(v.get(names[0], 0) for v in grp)
    work = [v for v in grp if v.get(names[0], 0) > 0.01 * ref]
    print("%3d %-46s %2d " % (sv, kn.replace("void tfq::", "")[:46], len(work)) + " ".join("%.5g" % (sum(v.get(c, 0) for v in work) / max(1, len(work))) for c in names))
