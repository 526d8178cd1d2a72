#!/usr/bin/env python3
"""Can the chip overlap the MFMA-bound multiply with an HBM stream of the size of the fused epilogue?
Runs the stand-alone multiply on one stream and elementwise traffic (5 S) on another, alone and together."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
pr, prec, desc = build_problem("fd2d_16x16_z", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
sol = T.Solver(s1.cuda_stream)
sol.create_plan(pr); view = sol.plan_view()
At = pr.A.transpose(0, 2, 1)
An = torch.from_numpy(np.ascontiguousarray(np.stack([At.real, At.imag], axis=1))).cuda()
Xn = torch.rand((pr.nnzbX, 2, 16, 16), dtype=torch.float64, device="cuda"); Yn = torch.empty_like(Xn)
dS = torch.from_numpy(view["starts"].view(np.int32)).cuda(); dP = torch.from_numpy(view["pairs"].view(np.int32)).cuda()
a, b, c, d, e = (torch.rand_like(Xn) for _ in range(5))
def mult():
    T.lib.tfqmrgpuExt_multiply(sol.handle, b"z", 16, 16, pr.nnzbX, dS.data_ptr(), dP.data_ptr(), An.data_ptr(), Xn.data_ptr(), Yn.data_ptr())
def stream():
    with torch.cuda.stream(s2):
        torch.add(a, b, out=c)      # 3 S
        torch.mul(d, 1.5, out=e)    # 2 S
def timed(fm, fs, reps=20):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1); s2.wait_event(e0)
    for _ in range(reps):
        if fm: mult()
        if fs: stream()
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
    e1.record(s1); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for _ in range(2):
    print("multiply alone %.3f ms | stream (5 S) alone %.3f ms | both concurrently %.3f ms" % (timed(True, False), timed(False, True), timed(True, True)))
