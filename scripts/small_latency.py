#!/usr/bin/env python3
"""wall time of repeated solves of a small system (launch/latency-bound regime)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from tfqmrgpu_amd.fd_generator import FDExample
for args in [(6, 24, 4, 2, -0.25, 4), (1.75, 6.75, 2, 3, 0.0, 4), (10, 60, 4, 2, 0.0, 4)]:
    pr = FDExample(*args).problem()
    stream = torch.cuda.Stream()
    with T.Solver(stream.cuda_stream) as s:
        s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, "z"))
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        s.solve(pr.tolerance, 2000)
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); st = s.solve(pr.tolerance, 2000); ts.append(time.perf_counter() - t0)
        it = s.get_info()["iterations"]
        print("FD %s: nnzbX %d, %d iterations, solve min %.3f ms median %.3f ms -> %.1f us/iteration" % (args, pr.nnzbX, it, min(ts)*1e3, sorted(ts)[5]*1e3, min(ts)/it*1e6))
