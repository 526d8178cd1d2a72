#!/usr/bin/env python3
"""Mixed precision ('m') against 'z' on one workload: time per solve, float iterations per refinement cycle, the double residual after every
cycle.  usage: python scripts/mixed_trace.py <workload> [threshold]   (TFQMRGPU_LIB / lab switches from the environment)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
pr, prec, desc = build_problem(sys.argv[1], 0)
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-9
out = {}
for p in ("z", "m"):
    with T.Solver() as s:
        s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, p))
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        st = s.solve(tol, 2000)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): st = s.solve(tol, 2000)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
        info = s.get_info()
        out[p] = ms
        line = "%-22s %s status %d iterations %3d residual %.2e  %.3f ms per solve" % (sys.argv[1], p, st, info["iterations"], info["residual"], ms)
        if p == "m":
            res, its = s.refinement_history(with_iterations=True)
            line += "  cycles " + " ".join("%d:%.1e" % (i, r) for i, r in zip(list(its), list(res))) + "  speedup %.3f" % (out["z"] / ms)
        print(line, flush=True)
