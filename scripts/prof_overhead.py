#!/usr/bin/env python3
"""What the per-kernel HIP events of bench.py's timed region cost: K solves with profiling (events + the host-side
queries, as bench.py runs them) against K solves without.  usage: python scripts/prof_overhead.py [workload] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfqmrgpu_amd as T
from bench import build_problem

name = sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pr, prec, desc = build_problem(name, 0)
s = T.Solver()
s.create_plan(pr)
s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
for _ in range(2):
    s.solve(pr.tolerance, 2000)

def run(mode):
    s.set_profiling(mode > 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        s.solve(pr.tolerance, 2000)
        if mode > 1:
            s.get_info(); s.profile(); s.profile(gated=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3

for rep in range(3):
    print("ms per solve: no events %.3f | events %.3f | events + queries (bench.py) %.3f" % (run(0), run(1), run(2)), flush=True)
print("iterations", s.get_info()["iterations"])
