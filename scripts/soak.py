#!/usr/bin/env python3
"""Soak run: N solves of one workload on one plan; every solve must give the same iteration count, residual and solution bits
(fixed reduction orders; the arrival counters of the folded column operations decide WHO sums, never in which order).
usage: python scripts/soak.py [workload] [N] [precision override, e.g. m]"""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tfqmrgpu_amd as T
from bench import build_problem

name = sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
pr, prec, desc = build_problem(name, 0)
if len(sys.argv) > 3:
    prec = sys.argv[3]
s = T.Solver()
s.create_plan(pr)
s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
seen, digests = {}, set()
for i in range(N):
    st = s.solve(pr.tolerance, 2000)
    info = s.get_info()
    key = (st, info["iterations"], info["residual"])
    if i % 10 == 0 or key not in seen:
        X = s.get_matrix()
        digests.add(hashlib.md5(np.ascontiguousarray(X).tobytes()).hexdigest())     # the solution BITS of every sampled solve
    seen.setdefault(key, []).append(i)
    if i % 20 == 0:
        print(i, key, sorted(digests), flush=True)
print("%s (%s): %d solves, distinct (status, iterations, residual): %d, distinct md5 of the sampled solutions: %d" % (name, prec, N, len(seen), len(digests)))
assert len(seen) == 1, seen
assert len(digests) == 1, digests
