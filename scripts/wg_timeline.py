#!/usr/bin/env python3
"""Timeline of one wave per work group of k_spmm_ilv16 in a (small) plain multiply: where the ~20 us of a latency-bound launch go.
Needs the stamps variant: scripts/build_variant.sh stamps -DTFQ_LAB -DTFQ_LAB_STAMPS (tfq_spmm.hip: TFQ_STAMP; 100 MHz clock = 10 ns ticks).
usage: python scripts/wg_timeline.py [workload ...]   (16 x 16 complex<double> workloads: bench.py names, st:16:16:z:..., FD:a,b,c,d,e,f)
WG_EPI=1 | 2: the LAST working launch of that fused multiply in a solve instead of a plain multiply (0)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("TFQMRGPU_LIB", os.path.join(ROOT, "scripts", "bin", "libtfQMRgpu_stamps.so"))
import numpy as np, torch
import tfqmrgpu_amd as T
from bench import build_problem
from tfqmrgpu_amd.fd_generator import FDExample
NAMES = ["entry", "state known", "chunk known", "pair range known", "operands landed", "products done", "stores issued", "records written / end"]
T.lib.tfqmrgpuLab_stamps.argtypes = [C.c_void_p, C.c_int]
EPI = int(os.environ.get("WG_EPI", "0"))
for name in (sys.argv[1:] or ["FD:6,24,4,2,-0.25,4", "st:16:16:z:8:8:1", "st:16:16:z:32:32:4"]):
    if name.startswith("FD:"):
        pr = FDExample(*[float(v) if "." in v else int(v) for v in name[3:].split(",")]).problem(); prec = "z"
    else:
        pr, prec, _ = build_problem(name, 0)
    with T.Solver() as s:
        s.create_plan(pr); s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        s.solve(pr.tolerance, 200)
        rec = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
        s.apply_operator(-3); torch.cuda.synchronize()
        assert T.lib.tfqmrgpuLab_stamps(rec.data_ptr(), EPI) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if EPI: s.solve(pr.tolerance, 200)
        else: s.apply_operator(-1)
        e1.record(); torch.cuda.synchronize()
        T.lib.tfqmrgpuLab_stamps(None, 0)
        st = rec.cpu().numpy()
        st = st[st[:, 0] > 0]
        t0 = st[:, 0].min()
        d = (st - st[:, :1]) * 0.01            # us since the work group's own entry
        print("%s (epilogue %d): %d work groups, %s (events) %.1f us, first entry -> last end %.1f us; work groups enter over %.1f us" % (
            name, EPI, len(st), "solve" if EPI else "launch", e0.elapsed_time(e1) * 1e3, (st[:, 7].max() - t0) * 0.01, (st[:, 0].max() - t0) * 0.01))
        for k in range(1, 8):
            print("   %-24s median %6.2f us after entry (step %5.2f)   90 %%: %6.2f" % (NAMES[k], np.median(d[:, k]), np.median(d[:, k] - d[:, k - 1]), np.percentile(d[:, k], 90)))
