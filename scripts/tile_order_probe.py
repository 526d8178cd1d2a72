#!/usr/bin/env python3
"""Does a 2-D tiled numbering of the block rows (so that a chunk of the solver is a tile, not a strip, of the grid)
speed the fused multiplies up?  Same stencil problem with raster and with tiled row numbers.
usage: python scripts/tile_order_probe.py [nx] [ncols] [tx] [ty] [LM] [LN]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tfqmrgpu_amd as T
from tfqmrgpu_amd import problems as PR

def permute_rows(pr, new_of_old):
    mb = pr.mb
    old_of_new = np.argsort(new_of_old)
    def csr(rp, ci, vals, permute_cols):
        rows = np.repeat(np.arange(mb), np.diff(rp))
        nr = new_of_old[rows]
        nc = new_of_old[ci] if permute_cols else ci
        o = np.lexsort((nc, nr))
        rp2 = np.zeros(mb + 1, np.int64); np.add.at(rp2, nr + 1, 1)
        return np.cumsum(rp2).astype(np.int32), nc[o].astype(np.int32), (None if vals is None else vals[o])
    rpA, ciA, A = csr(pr.rowPtrA, pr.colIndA, pr.A, True)
    rpX, ciX, _ = csr(pr.rowPtrX, pr.colIndX, None, False)
    rpB, ciB, B = csr(pr.rowPtrB, pr.colIndB, pr.B, False)
    return T.Problem(rpA, ciA, A, rpX, ciX, rpB, ciB, B, None, pr.tolerance)

nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ncols = int(sys.argv[2]) if len(sys.argv) > 2 else 48
tx = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ty = int(sys.argv[4]) if len(sys.argv) > 4 else 2
LM = int(sys.argv[5]) if len(sys.argv) > 5 else 16
LN = int(sys.argv[6]) if len(sys.argv) > 6 else LM
pr = PR.stencil_2d(nx, nx, LM, LN, ncols, seed=7)
y, x = np.divmod(np.arange(nx * nx), nx)
key = ((y // ty) * (nx // tx) + (x // tx)) * (tx * ty) + (y % ty) * tx + (x % tx)
tiled = permute_rows(pr, np.argsort(np.argsort(key)))
for name, p in (("raster", pr), ("tiled %dx%d" % (tx, ty), tiled), ("raster", pr)):
    with T.Solver() as s:
        s.create_plan(p); s.set_buffer(nbytes=s.buffer_size(LM, LN, "z"))
        s.set_matrix("A", p.A); s.set_matrix("B", p.B)
        s.set_profiling(True)
        tot = {}
        for _ in range(3):
            st = s.solve(1e-9, 200)
            for k, (n, ms) in s.profile().items():
                a = tot.setdefault(k, [0, 0.0]); a[0] += n; a[1] += ms
        info = s.get_info()
        print("%-10s status %d iterations %d residual %.2e | spmm_v4_dot %.4f ms  spmm_v5_nrm_dot %.4f ms  x_v6_v7 %.4f ms" % (
            name, st, info["iterations"], info["residual"], tot["spmm_v4_dot"][1] / tot["spmm_v4_dot"][0],
            tot["spmm_v5_nrm_dot"][1] / tot["spmm_v5_nrm_dot"][0], tot["x_v6_v7"][1] / tot["x_v6_v7"][0]))
