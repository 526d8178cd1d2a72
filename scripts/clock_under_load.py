#!/usr/bin/env python3
"""Shader clock while the fused multiplies of a solve run (lab build: tfqmrgpuLab_clockRecord, tfq_spmm.hip): average frequency over all
work groups of k_spmm_ilv16 = clocks / ticks * 100 MHz, and the average lifetime of a work group.  usage: python scripts/clock_under_load.py [workload]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# needs a build with the clock hook: scripts/build_variant.sh clock -DTFQ_LAB -DTFQ_LAB_CLOCK
os.environ.setdefault("TFQMRGPU_LIB", os.path.join(ROOT, "scripts", "bin", "libtfQMRgpu_clock.so"))
import torch
import tfqmrgpu_amd as T
from bench import build_problem
pr, prec, desc = build_problem(sys.argv[1] if len(sys.argv) > 1 else "fd2d_16x16_z", 0)
rec = torch.zeros(3, dtype=torch.int64, device="cuda")
T.lib.tfqmrgpuLab_clockRecord.argtypes = [C.c_void_p]
with T.Solver() as s:
    s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
    s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
    s.solve(pr.tolerance, 2000)
    assert T.lib.tfqmrgpuLab_clockRecord(rec.data_ptr()) == 0
    for what in ("solve", "multiply"):
        rec.zero_(); torch.cuda.synchronize()
        if what == "solve":
            s.solve(pr.tolerance, 2000)
        else:
            s.apply_operator(-20)
        torch.cuda.synchronize()
        c, w, n = (int(v) for v in rec.cpu())
        print("%-8s: %d work groups of k_spmm_ilv16, shader clock %.3f GHz (lifetimes are inflated by the record's own atomics: not reported)" % (
            what, n, c / max(1, w) * 0.1))
    T.lib.tfqmrgpuLab_clockRecord(None)
