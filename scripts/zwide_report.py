#!/usr/bin/env python3
"""Deviation of the HIP path from the oracle on the wide complex<double> fixtures (tests/test_gpu_hash_mode.py: Z_WIDE), four- and three-product
form: iteration counts, bound history (whole | first half), residual, solution.  The tolerances of the test sit above what this prints on MI355X."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tfqmrgpu_amd as T
from oracle import pyoracle as oracle
from test_gpu_hash_mode import Z_WIDE
for name in sorted(Z_WIDE):
    pr = Z_WIDE[name]()
    for three in (0, 1):
        st, X, info = T.solve_problem(pr, "z", threshold=pr.tolerance, max_iterations=300, three_products=bool(three))
        st0, X0, info0 = oracle.solve(pr, "z", threshold=pr.tolerance, max_iterations=300, v3=T.hash_shadow_vector(pr).reshape(-1))
        h, h0 = np.asarray(info["bound_history"]), np.asarray(info0["bound_history"])
        n = min(len(h), len(h0)); half = (len(h0) + 1) // 2
        print("%-16s three_products %d status %d %d iterations %d %d history %.1e first half %.1e residual %.1e solution %.1e" % (
            name, three, st, st0, info["iterations"], info0["iterations"], np.abs(h[:n] / h0[:n] - 1).max(), np.abs(h[:half] / h0[:half] - 1).max(),
            abs(info["residual"] / info0["residual"] - 1), np.abs(X - X0).max() / np.abs(X0).max()), flush=True)
