#!/usr/bin/env python3
"""Does the numbering of the block rows matter?  The 5-point block stencil of BASELINE config 5 (8x8 z, 256 x 256 grid, 8 block columns) once with
the grid points numbered line by line (as bench.py builds it) and once tile by tile (T x T points per tile: the +-nx neighbours of a row are then
mostly inside its own chunk).  The library is unchanged: only the caller's row numbers differ.  usage: python scripts/tiled_rows_probe.py [T] [LM]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tfqmrgpu_amd as T_
from tfqmrgpu_amd import problems as PR

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
LM = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nx = ny = 256 if LM == 8 else 128
ncols = 8 if LM == 8 else 32


def permuted(pr, ident):
    """the same system with block row r renamed ident[r] (rows AND columns of A, rows of X and B)"""
    mb = pr.mb
    inv = np.argsort(ident)                      # new row n was old row inv[n]
    def rows_of(rp): return np.repeat(np.arange(mb), np.diff(rp))
    out = []
    for rp, ci, vals, is_a in ((pr.rowPtrA, pr.colIndA, pr.A, True), (pr.rowPtrX, pr.colIndX, None, False), (pr.rowPtrB, pr.colIndB, pr.B, False)):
        r = ident[rows_of(rp)]
        c = ident[ci] if is_a else ci
        order = np.lexsort((c, r))
        nrp = np.zeros(mb + 1, np.int64); np.add.at(nrp, r + 1, 1); nrp = np.cumsum(nrp)
        out.append((nrp.astype(np.int32), np.asarray(c)[order].astype(np.int32), None if vals is None else vals[order]))
    (rpA, ciA, A), (rpX, ciX, _), (rpB, ciB, B) = out
    return T_.Problem(rpA, ciA, A, rpX, ciX, rpB, ciB, B, None, pr.tolerance)


def run(tag, pr):
    s = T_.Solver()
    s.create_plan(pr)
    s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, "z"))
    s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
    s.solve(pr.tolerance, 2000)
    s.set_profiling(1)
    acc = {}
    for _ in range(3):
        st = s.solve(pr.tolerance, 2000)
        for k, (n, ms) in s.profile().items():
            a = acc.setdefault(k, [0, 0.0]); a[0] += n; a[1] += ms
    info = s.get_info()
    it = sum(v[1] / v[0] for k, v in acc.items() if k != "probe" and v[0])
    print("%-22s status %d iterations %d | spmm_v4_dot %.4f spmm_v5_nrm_dot %.4f x_v6_v7 %.4f probe %.4f | iteration %.4f ms" % (
        tag, st, info["iterations"], acc["spmm_v4_dot"][1] / acc["spmm_v4_dot"][0], acc["spmm_v5_nrm_dot"][1] / acc["spmm_v5_nrm_dot"][0],
        acc["x_v6_v7"][1] / acc["x_v6_v7"][0], acc["probe"][1] / max(1, acc["probe"][0]), it), flush=True)
    s.close()


pr = PR.stencil_2d(nx, ny, LM, LM, ncols, seed=5)
run("line by line", pr)
x, y = np.meshgrid(np.arange(nx), np.arange(ny))
x, y = x.reshape(-1), y.reshape(-1)             # old row r = y * nx + x
for t in (T, 2 * T):
    ident = (((y // t) * (nx // t) + x // t) * t * t + (y % t) * t + x % t).astype(np.int64)
    run("tiles of %d x %d points" % (t, t), permuted(pr, ident))
