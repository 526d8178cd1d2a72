#!/usr/bin/env python3
"""r04: what setMatrix / getMatrix cost a caller whose arrays are pageable host memory (the reference's API hands over host pointers):
seconds and GB/s of getMatrix('X'), setMatrix('X') and setMatrix('B') on one workload.  usage: python scripts/host_arrays.py [workload] [reps]
(A/B: TFQMRGPU_LIB=<lab build> TFQMRGPU_PIPED_COPY=0|1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem

name = sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
pr, prec, desc = build_problem(name, 0)
s = T.Solver()
s.create_plan(pr)
s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
st = s.solve(pr.tolerance, 2000)
cdt = np.complex128 if prec == "z" else np.complex64
X = s.get_matrix().astype(cdt)
def best(f):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts)
out = np.zeros((pr.nnzbX, pr.LM, pr.LN, 2), dtype=np.float64 if prec == "z" else np.float32)
def get():
    T._check(T.lib.tfqmrgpu_bsrsv_getMatrix(s.handle, s.plan, b"X", T._ptr(out), prec.encode(), pr.LN, pr.LM, b"n", T.LAYOUT_RIRIRIRI), "get")
tg = best(get)
Xc = np.ascontiguousarray(X)
def setx():
    T._check(T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"X", T._ptr(Xc), prec.encode(), pr.LN, pr.LM, b"n", T.LAYOUT_RIRIRIRI), "set")
tsx = best(setx)
Bc = np.ascontiguousarray(pr.B.astype(cdt))
def setb():
    T._check(T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"B", T._ptr(Bc), prec.encode(), pr.LN, pr.LM, b"n", T.LAYOUT_RIRIRIRI), "setB")
tsb = best(setb)
ok = np.array_equal(out[..., 0] + 1j * out[..., 1], X)
print("%s piped=%s  X %.0f MB: getMatrix %.1f ms (%.1f GB/s)  setMatrix %.1f ms (%.1f GB/s) | B %.1f MB: setMatrix %.2f ms | round trip exact: %s" % (
    name, os.environ.get("TFQMRGPU_PIPED_COPY", "default"), Xc.nbytes / 1e6, tg * 1e3, Xc.nbytes / tg / 1e9, tsx * 1e3, Xc.nbytes / tsx / 1e9, Bc.nbytes / 1e6, tsb * 1e3, ok))
