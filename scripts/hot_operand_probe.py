#!/usr/bin/env python3
"""How fast is a multiply kernel when its operands are cache-hot?  Times tfqmrgpuExt_multiply with the real pair list of a
workload, with every product reading A block 0 / X block 0 (pure pipe + issue rate of the kernel), and with only A or only X hot.
usage: python scripts/hot_operand_probe.py [workload] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
name = sys.argv[1] if len(sys.argv) > 1 else "stencil3d_32x32_c"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pr, prec, desc = build_problem(name, 0)
s = T.Solver(); s.create_plan(pr); v = s.plan_view()
real = torch.float64 if prec == "z" else torch.float32
A = torch.rand((pr.nnzbA, 2, pr.LM, pr.LM), dtype=real, device="cuda") - 0.5
X = torch.rand((pr.nnzbX, 2, pr.LM, pr.LN), dtype=real, device="cuda") - 0.5
Y = torch.empty_like(X)
dS = torch.from_numpy(v["starts"].view(np.int32)).cuda()
flops = v["nPairs"] * 8.0 * pr.LM * pr.LM * pr.LN
peak = 78.6 if prec == "z" else 157.3
def run(tag, pairs):
    dP = torch.from_numpy(np.ascontiguousarray(pairs).view(np.int32)).cuda()
    f = lambda: T._check(T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(), A.data_ptr(), X.data_ptr(), Y.data_ptr()), "m")
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / reps)
    ms = min(ts)
    print("%-28s %.4f ms  %.1f TFLOP/s  (%.3f of %.1f)" % (tag, ms, flops / ms * 1e-9, flops / ms * 1e-9 / peak, peak), flush=True)
p = v["pairs"].copy().reshape(-1, 2)
print("#", desc, "pairs", v["nPairs"])
run("real pair list", p)
q = p.copy(); q[:, 0] = 0; run("A hot (block 0)", q)
q = p.copy(); q[:, 1] = 0; run("X hot (block 0)", q)
q = p.copy(); q[:, :] = 0; run("A and X hot", q)
q = p.copy(); q[:, 0] %= 64; q[:, 1] %= 64; run("A, X from 64 hot blocks", q)
