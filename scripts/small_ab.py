#!/usr/bin/env python3
"""Same-box A/B of whole library builds on SMALL systems (latency-bound): time per iteration of repeated solves, each build in a process of its own
(TFQMRGPU_LIB).  usage: python scripts/small_ab.py lib1 [lib2 ...] -- workload [workload ...]   (workloads: bench.py names, st:..., FD:a,b,c,d,e,f)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
from tfqmrgpu_amd.fd_generator import FDExample
name = sys.argv[1]
if name.startswith("FD:"):
    pr = FDExample(*[float(v) if "." in v else int(v) for v in name[3:].split(",")]).problem(); prec = "z"
else:
    pr, prec, _ = build_problem(name, 0)
with T.Solver() as s:
    s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
    s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
    s.solve(pr.tolerance, 200)
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = s.solve(pr.tolerance, 200); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    info = s.get_info()
    it = info["iterations"] if st == 0 else 200
    print("%%-26s %%-28s nnzbX %%6d status %%d iterations %%3d residual %%.3e  %%.1f us per iteration (median %%.1f)" %% (
        os.path.basename(os.environ.get("TFQMRGPU_LIB", "default")), name, pr.nnzbX, st, it, info["residual"], min(ts) / it * 1e6, sorted(ts)[10] / it * 1e6), flush=True)
''' % ROOT
libs, wls = sys.argv[1:sys.argv.index("--")], sys.argv[sys.argv.index("--") + 1:]
for wl in wls:
    for lib in libs:
        subprocess.call([sys.executable, "-c", CHILD, wl], env=dict(os.environ, TFQMRGPU_LIB=os.path.join(ROOT, lib)))
