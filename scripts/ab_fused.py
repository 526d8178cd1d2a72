#!/usr/bin/env python3
"""Same-box A/B of whole library builds on one workload: per-launch time of the two fused multiplies, the plain multiply on the
plan's data and the iteration, each build in a process of its own (TFQMRGPU_LIB).  usage: python scripts/ab_fused.py <workload> lib1 lib2 ...
(AB_MAXIT=n caps the iterations: timing-only probe builds give wrong results and would not stop on their own)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
pr, prec, desc = build_problem(sys.argv[1], 0)
MAXIT = int(os.environ.get('AB_MAXIT', 2000))
s = T.Solver()
s.create_plan(pr)
s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
s.solve(pr.tolerance, MAXIT)
s.set_profiling(1)
acc = {}
for _ in range(3):
    st = s.solve(pr.tolerance, MAXIT)
    first = s.profile(first=True)
    for k, (n, ms) in s.profile().items():
        a = acc.setdefault(k, [0, 0.0]); a[0] += n - first[k][0]; a[1] += ms - first[k][1]
s.set_profiling(0)
info = s.get_info()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): s.solve(pr.tolerance, MAXIT)
torch.cuda.synchronize(); solve_ms = (time.perf_counter() - t0) / 5 * 1e3
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.apply_operator(20); torch.cuda.synchronize()
e0.record(); s.apply_operator(20); e1.record(); torch.cuda.synchronize()
it = sum(v[1] / v[0] for k, v in acc.items() if k != "probe" and v[0])
g = lambda k: acc[k][1] / max(1, acc[k][0])
print("%%-44s st %%d it %%d res %%.3e | spmm_v4_dot %%.4f spmm_v5_nrm_dot %%.4f x_v6_v7 %%.4f xpay %%.4f v5_nrm %%.4f | multiply %%.4f | iteration %%.4f ms | solve %%.3f ms" %% (
    os.environ.get("AB_TAG", "default"), st, info["iterations"], info["residual"], g("spmm_v4_dot"), g("spmm_v5_nrm_dot"), g("x_v6_v7"), g("xpay_v6"), g("v5_nrm"),
    e0.elapsed_time(e1) / 20, it, solve_ms), flush=True)
if os.environ.get("AB_ALL"):
    print("    " + " ".join("%%s %%.4f" %% (k, g(k)) for k in acc), flush=True)
s.close()
''' % ROOT
wl = sys.argv[1]
for lib in sys.argv[2:]:
    env = dict(os.environ)
    lib, _, switches = lib.partition("@")          # lib@TFQMRGPU_X=1,TFQMRGPU_Y=2: tuning switches of a LAB build (tfq_switch.hpp)
    if lib == "lab":
        lib = "tfqmrgpu_amd/lib/libtfQMRgpu_lab.so"
    if lib != "default":
        env["TFQMRGPU_LIB"] = os.path.join(ROOT, lib)
    for kv in filter(None, switches.split(",")):
        k, _, v = kv.partition("="); env[k] = v
    env["AB_TAG"] = os.path.basename(lib) + ("@" + switches if switches else "")
    subprocess.call([sys.executable, "-c", CHILD, wl], env=env)
