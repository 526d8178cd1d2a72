#!/usr/bin/env python3
"""a few solves of one small system (for kernel traces): python scripts/small_one.py [rsb rtb be dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tfqmrgpu_amd as T
from tfqmrgpu_amd.fd_generator import FDExample
a = [float(v) for v in sys.argv[1:5]] if len(sys.argv) > 4 else [6, 24, 4, 2]
pr = FDExample(a[0], a[1], int(a[2]), int(a[3]), -0.25 if a[3] == 2 else 0.0, 4).problem()
with T.Solver() as s:
    s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, "z"))
    s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
    for _ in range(4):
        st = s.solve(pr.tolerance, 2000)
    print(st, s.get_info())
