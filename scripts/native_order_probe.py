#!/usr/bin/env python3
"""r04: the stand-alone multiply on the caller's listing with a PREPARED launch order (tfqmrgpuExt_multiplyPrepare, tfq_order.cpp): mode 0 = caller's
order, 1 = (column group, row band, column, row), 2 = the same + heavy work groups first.  BASELINE config 1's plan file in `f` and `z`, and P2's
native listing.  ms per launch by HIP events over batches of launches back to back, TFLOP/s, and Y bit-identical with mode 0.
usage: python scripts/native_order_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np, torch
import tfqmrgpu_amd as T
from tfqmrgpu_amd.bench_tfqmrgpu import _read_plan
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50


def run(tag, prec, lm, ln, nY, nA, nX, starts, pairs):
    real = torch.float64 if prec == "z" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(5)
    dA = torch.rand((nA, 2, lm, lm), dtype=real, device="cuda", generator=g) * 2 - 1
    dX = torch.rand((nX, 2, lm, ln), dtype=real, device="cuda", generator=g) * 2 - 1
    dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
    s = T.Solver()
    nPairs = len(pairs) // 2
    flop = nPairs * 8.0 * lm * lm * ln
    ref = None
    for mode in (0, 1, 3, 4, 0, 1, 3, 4):
        dY = torch.zeros((nY, 2, lm, ln), dtype=real, device="cuda")
        order = C.c_void_p(None)
        T._check(T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, prec.encode(), lm, ln, nY, dS.data_ptr(), dP.data_ptr(), mode, C.byref(order)), "prepare")
        call = (s.handle, prec.encode(), lm, ln, nY, dS.data_ptr(), dP.data_ptr(), dA.data_ptr(), dX.data_ptr(), dY.data_ptr(), order)
        fn = T.lib.tfqmrgpuExt_multiplyOrdered
        for _ in range(5): T._check(fn(*call), "multiply")
        ms = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(reps): fn(*call)
            e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1) / reps)
        m = sorted(ms)[1]
        if ref is None: ref = dY.clone()
        same = bool(torch.equal(ref, dY))
        T.lib.tfqmrgpuExt_multiplyRelease(order)
        print("%-26s %s mode %d: %.4f ms per launch = %6.1f TFLOP/s  (Y %s)" % (tag, prec, mode, m, flop / m / 1e9, "bit-identical with mode 0" if same else "DIFFERENT"), flush=True)
    s.close()


nY, nA, nX, starts, pairs = _read_plan(os.path.join(ROOT, "tests", "golden", "plan_unordered.14-287-16.gz"))
run("config 1 plan file", "c", 16, 16, nY, nA, nX, starts, pairs)
run("config 1 plan file", "z", 16, 16, nY, nA, nX, starts, pairs)
from bench import build_problem
pr, prec, desc = build_problem("fd2d_16x16_z", 0)
s = T.Solver(); s.create_plan(pr); view = s.plan_view(); s.close()
run("P2 native listing", "z", 16, 16, pr.nnzbX, pr.nnzbA, pr.nnzbX, view["starts"].astype(np.uint32), view["pairs"].astype(np.uint32))
pr, prec, desc = build_problem("st:16:16:c:96:96:16", 0)
s = T.Solver(); s.create_plan(pr); view = s.plan_view(); s.close()
run("16x16 c stencil 96^2 x 16", "c", 16, 16, pr.nnzbX, pr.nnzbA, pr.nnzbX, view["starts"].astype(np.uint32), view["pairs"].astype(np.uint32))
