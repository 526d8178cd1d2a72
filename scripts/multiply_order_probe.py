#!/usr/bin/env python3
"""Stand-alone multiply (tfqmrgpuExt_multiply) with the Y blocks in the caller's row-major order against the same
products listed column by column (the order the solver uses internally), optionally dealt to the XCDs in contiguous parts.
usage: python scripts/multiply_order_probe.py [workload] [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem, kernel_model, roof

name = sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pr, prec, desc = build_problem(name, 0)
s = T.Solver(); s.create_plan(pr)
view = s.plan_view()
model = kernel_model(pr, prec, view["nPairs"], len(np.unique(view["pairs"][0::2])))
real = torch.float64 if prec == "z" else torch.float32
nA = len(pr.A)
An = (torch.rand((nA, 2, pr.LM, pr.LM), dtype=real, device="cuda") * 2 - 1)
Xn = torch.rand((pr.nnzbX, 2, pr.LM, pr.LN), dtype=real, device="cuda") * 2 - 1
Yn = torch.empty_like(Xn)
starts, pairs, col = view["starts"].astype(np.int64), view["pairs"].reshape(-1, 2), view["colindx"].astype(np.int64)
row = np.repeat(np.arange(len(pr.rowPtrX) - 1), np.diff(np.asarray(pr.rowPtrX, dtype=np.int64)))

def regroup(perm):
    """pair list with Y block n' = perm[n'] of the original; Y is written in the new order"""
    lens = np.diff(starts)[perm]
    ns = np.concatenate([[0], np.cumsum(lens)])
    idx = np.concatenate([np.arange(starts[p], starts[p + 1]) for p in perm]) if len(perm) else np.zeros(0, np.int64)
    return ns.astype(np.uint32), np.ascontiguousarray(pairs[idx].reshape(-1).astype(np.uint32))

def time_it(st, pa, label):
    dS = torch.from_numpy(st.view(np.int32)).cuda(); dP = torch.from_numpy(pa.view(np.int32)).cuda()
    call = lambda: T._check(T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(), An.data_ptr(), Xn.data_ptr(), Yn.data_ptr()), "mult")
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(5):
        e0.record()
        for _ in range(reps): call()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    print("%-46s min %.4f ms -> %s" % (label, min(ts), json.dumps(roof(model["multiply"][0], model["multiply"][1], min(ts), prec))))

n = pr.nnzbX
time_it(*regroup(np.arange(n)), "row-major (caller's order)")
colmajor = np.lexsort((row, col))
time_it(*regroup(colmajor), "column-major")
# groups of 8 columns, bands of 4 rows inside (the sort key of the solver's launch order), no XCD dealing
key = np.lexsort((col, row // 4, col // 8))
time_it(*regroup(key), "(8 columns, band of 4 rows, column)")
# the same dealt to the XCDs: work group b (4 blocks) -> XCD b % 8 gets a contiguous eighth of the list
for base, label in ((colmajor, "column-major"), (key, "(8 columns, band, column)")):
    nwg = (n + 3) // 4
    wg = np.arange(nwg)
    q, r = divmod(nwg, 8)
    begin = np.concatenate([[0], np.cumsum([q + (1 if x < r else 0) for x in range(8)])])
    src = np.empty(nwg, np.int64); w = 0
    for i in range(q + 1):
        for x in range(8):
            if begin[x] + i < begin[x + 1]:
                src[w] = begin[x] + i; w += 1
    blocks = (src[:, None] * 4 + np.arange(4)[None, :]).reshape(-1)
    blocks = blocks[blocks < n]
    time_it(*regroup(base[blocks]), label + ", XCD x <- contiguous eighth")
s.close()
