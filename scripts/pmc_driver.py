#!/usr/bin/env python3
"""Small driver for rocprofv3 --pmc runs: one solve with a few iterations + a few stand-alone multiplies
of the bench workload.  usage: python3 scripts/pmc_driver.py [workload] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
name = sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pr, prec, desc = build_problem(name, 0)
s = T.Solver()
s.create_plan(pr)
view = s.plan_view()
s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
st = s.solve(1e-30, iters)
s.set_matrix("X", np.random.default_rng(1).uniform(-1, 1, (pr.nnzbX, pr.LM, pr.LN)) + 0j)
s.apply_operator(3)      # the multiply on the plan's data (solver's kernel and element order, EPI_NONE)
real = torch.float64 if prec == "z" else torch.float32
At = pr.A.transpose(0, 2, 1)
An = torch.from_numpy(np.ascontiguousarray(np.stack([At.real, At.imag], axis=1))).to(real).cuda()
Xn = torch.rand((pr.nnzbX, 2, pr.LM, pr.LN), dtype=real, device="cuda") * 2 - 1
Yn = torch.empty_like(Xn)
dS = torch.from_numpy(view["starts"].view(np.int32)).cuda(); dP = torch.from_numpy(view["pairs"].view(np.int32)).cuda()
for _ in range(3):
    T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), pr.LM, pr.LN, pr.nnzbX, dS.data_ptr(), dP.data_ptr(), An.data_ptr(), Xn.data_ptr(), Yn.data_ptr())
torch.cuda.synchronize()
s.close()
print("pmc_driver done", st)
