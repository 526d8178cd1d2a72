#!/usr/bin/env python3
"""Where folding the column operations into the producers' tails stops paying: time per iteration of systems of growing chunk count with
the lab build's TFQMRGPU_FOLD_MAX = 0 (never fold) and = 1 << 30 (always), each in a process of its own.  FOLD_STREAM=created: on a stream of its own instead of the null stream.
usage: python scripts/fold_crossover.py  (needs tfqmrgpu_amd/lib/libtfQMRgpu_lab.so)"""
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
stream = torch.cuda.Stream() if os.environ.get('FOLD_STREAM') == 'created' else None
with T.Solver(stream.cuda_stream if stream else None) as s:
    s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
    s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
    s.solve(pr.tolerance, 200)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); st = s.solve(pr.tolerance, 200); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    info = s.get_info()
    it = info["iterations"] if st == 0 else 200
    print("%%-8s %%-28s fold_max %%-10s nnzbX %%6d status %%d iterations %%3d  %%.1f us per iteration" %% (os.environ.get("FOLD_STREAM", "null"), name, os.environ.get("TFQMRGPU_FOLD_MAX"), pr.nnzbX, st, it, min(ts) / it * 1e6), flush=True)
''' % ROOT
lab = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "libtfQMRgpu_lab.so")
for wl in (sys.argv[1:] or ["FD:1.75,6.75,2,3,0.0,4", "FD:6,24,4,2,-0.25,4", "st:16:16:z:24:24:4", "st:16:16:z:32:32:4", "st:16:16:z:48:48:4", "st:16:16:z:64:64:4", "st:16:16:c:48:48:4", "st:32:32:c:24:24:2", "st:8:8:z:64:64:4"]):
    for fm in ("0", str(1 << 30)):
        subprocess.call([sys.executable, "-c", CHILD, wl], env=dict(os.environ, TFQMRGPU_LIB=lab, TFQMRGPU_FOLD_MAX=fm))
