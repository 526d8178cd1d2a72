#!/usr/bin/env python3
"""Observed deviations of the HIP path from the CPU oracle, per fixture: what the tolerances written into
tests/test_gpu_parity.py are derived from (2 x the worst deviation seen).  Run on the GPU box:
    python tests/parity_report.py > gpurun_out/parity_report.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tfqmrgpu_amd as T  # noqa: E402
from conftest import ALL_NAMES, golden_solves, load_golden, load_problem  # noqa: E402
from oracle import pyoracle as O  # noqa: E402
from tfqmrgpu_amd import problems as PR  # noqa: E402

O.lib()


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    n = min(len(a), len(b))
    return np.abs(a[:n] - b[:n]) / np.maximum(np.abs(b[:n]), 1e-300)


def line(tag, st, info, st0, info0, X, X0):
    h, h0 = info["bound_history"], info0["bound_history"]
    r = rel(h, h0)
    half = (len(h0) + 1) // 2
    print("%-44s st %d/%d it %3d/%3d  hist relmax all %.2e first-half %.2e  res %.3e/%.3e rel %.2e  dX/max|X| %.2e" % (
        tag, st, st0, info["iterations"], info0["iterations"], r.max() if len(r) else 0, r[:half].max() if len(r) else 0,
        info["residual"], info0["residual"], abs(info["residual"] - info0["residual"]) / max(info0["residual"], 1e-300),
        np.abs(X - X0).max() / max(np.abs(X0).max(), 1e-300)), flush=True)


print("== glibc shadow vector (the reference CPU path's), golden fixtures")
for name in ALL_NAMES:
    pr, g = load_problem(name), load_golden(name)
    for prec, tol, maxit in golden_solves(g):
        st, X, info = T.solve_problem(pr, prec, threshold=tol, max_iterations=maxit, shadow_mode=T.SHADOW_GLIBC_RAND)
        st0, X0, info0 = O.solve(pr, prec, threshold=tol, max_iterations=maxit)
        line("%s %s tol %.0e" % (name, prec, tol), st, info, st0, info0, X, X0)

print("== hash shadow vector (the default, the benchmarked kernels), oracle fed with the same vector")
import test_gpu_hash_mode as H  # noqa: E402
cases = [(n, load_problem(n)) for n in ALL_NAMES]
cases += [(n, make()) for n, make in sorted(H.CASES.items()) if n not in ALL_NAMES]     # the synthetic cases of the hash-mode tests
cases += [("cfg3 small 13-point 32x32", PR.stencil_2d(8, 8, 32, 32, 2, seed=3, points=13)),
          ("st 8x8 12x12 4 columns", PR.stencil_2d(12, 12, 8, 8, 4, seed=5))]
for name, pr in cases:
    for prec in "zc":
        tol = pr.tolerance if prec == "z" else 1e-4
        if name == "julia_kat":
            tol = 1.2e-8 if prec == "z" else 1.2e-5
        if name == "fd_8x8_3d" and prec == "c":
            tol = 1e-2
        v3 = T.hash_shadow_vector(pr)
        st, X, info = T.solve_problem(pr, prec, threshold=tol, max_iterations=300)
        st0, X0, info0 = O.solve(pr, prec, threshold=tol, max_iterations=300, v3=v3.reshape(-1))
        line("%s %s tol %.0e (hash)" % (name, prec, tol), st, info, st0, info0, X, X0)

print("== work vectors after exactly k iterations (tests/test_gpu_hash_mode.py::test_work_vectors_after_k_iterations_match_the_oracle):")
print("   max |v - v_oracle| / max |v_oracle| per vector (1 = x, 4 ... 9 = v4 ... v9)")
for name, prec, _ in H.STATE_CASES:  # every k, also those the test skips
    pr = H.CASES[name]()
    v3 = T.hash_shadow_vector(pr).reshape(-1)
    for k in H.STATE_ITERATIONS:
        st0, X0, info0 = O.solve(pr, prec, threshold=1e-30, max_iterations=k, v3=v3, dump_iteration=k)
        with T.Solver() as s:
            s.create_plan(pr)
            s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
            s.set_matrix("A", pr.A, "n")
            s.set_matrix("B", pr.B, "n")
            st = s.solve(1e-30, k)
            got = {w: s.get_work_vector(w) for w in (1, 4, 5, 6, 7, 8, 9)}
        worst = {w: np.abs(got[w] - v).max() / np.abs(v).max() for w, v in info0["vectors"].items()}
        print("%-16s %s k=%d st %d/%d  " % (name, prec, k, st, st0) + "  ".join("v%d %.1e" % (w, e) for w, e in worst.items()), flush=True)
