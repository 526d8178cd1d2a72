#!/usr/bin/env python3
"""Times the stand-alone BSR multiply (tfqmrgpuExt_multiply) and the fused solver kernels on one workload.
usage: python scripts/bench_multiply.py [workload] [reps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem, kernel_model, roof

name = sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
pr, prec, desc = build_problem(name, 0)
s = T.Solver()
s.create_plan(pr)
view = s.plan_view()
nbytes = s.buffer_size(pr.LM, pr.LN, prec)
s.set_buffer(nbytes=nbytes)
s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
model = kernel_model(pr, prec, view["nPairs"], len(np.unique(view["pairs"][0::2])))
real = torch.float64 if prec == "z" else torch.float32
At = pr.A.transpose(0, 2, 1)
An = torch.from_numpy(np.ascontiguousarray(np.stack([At.real, At.imag], axis=1))).to(real).cuda()
Xn = torch.rand((pr.nnzbX, 2, pr.LM, pr.LN), dtype=real, device="cuda") * 2 - 1
Yn = torch.empty_like(Xn)
dS = torch.from_numpy(view["starts"].view(np.int32)).cuda(); dP = torch.from_numpy(view["pairs"].view(np.int32)).cuda()
def mult():
    T._check(T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(), An.data_ptr(), Xn.data_ptr(), Yn.data_ptr()), "mult")
for _ in range(3): mult()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for r in range(5):
    e0.record()
    for _ in range(reps): mult()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / reps)
ms = min(ts)
print("multiply %s: min %.4f ms median %.4f ms -> %s" % (name, ms, sorted(ts)[len(ts)//2], json.dumps(roof(model["multiply"][0], model["multiply"][1], ms, prec))))
s.set_profiling(True)
tot = {}
for _ in range(3):
    st = s.solve(pr.tolerance, 2000)
    for k, (n, m) in s.profile().items():
        a = tot.setdefault(k, [0, 0.0]); a[0] += n; a[1] += m
info = s.get_info()
print("solve status %d iterations %d residual %.3e" % (st, info["iterations"], info["residual"]))
it_ms = 0
for k, (n, m) in tot.items():
    if n:
        line = "  %-16s %4d launches avg %.4f ms" % (k, n, m / n)
        if k in model:
            line += "  " + json.dumps(roof(model[k][0], model[k][1], m / n, prec))
        if k != "probe": it_ms += m / n
        print(line)
print("  per iteration %.4f ms" % it_ms)
s.close()
