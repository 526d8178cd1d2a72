#!/usr/bin/env python3
"""How much of the fused multiplies' operand re-fetch is a matter of the ORDER of the block rows and of the column-group size of
the launch order (TFQMRGPU_ORDER_G)?  The bench system P2 (generate_FD_example 16 120 4 2 -0.25) with its rows renumbered on the
caller's side (the library is unchanged): natural (packed-coordinate order of the generator), raster, strips of w grid columns walked
line by line, T x T tiles, Morton order.  Prints the per-launch times of the two fused multiplies and of the whole iteration.
usage: python scripts/row_order_probe.py [G list, e.g. 4,8,16] [orders, e.g. natural,strip8]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# TFQMRGPU_ORDER_G is a tuning switch: only the LAB build reads it (the product has its switches frozen since commit ab30726, tfq_switch.hpp)
os.environ.setdefault("TFQMRGPU_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tfqmrgpu_amd", "lib", "libtfQMRgpu_lab.so"))
import numpy as np
import tfqmrgpu_amd as T_
assert "lab" in os.path.basename(T_.LIB_PATH), "the column-group sweep needs the lab build: " + T_.LIB_PATH
from tfqmrgpu_amd.fd_generator import FDExample


def permuted(pr, ident):
    """the same system with block row r renamed ident[r] (rows AND columns of A, rows of X and B)"""
    mb = pr.mb
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


def morton(x, y):
    k = np.zeros(len(x), np.int64)
    for b in range(8):
        k |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return k


def run(tag, pr, G):
    os.environ["TFQMRGPU_ORDER_G"] = str(G)
    s = T_.Solver()
    s.create_plan(pr)
    s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, "z"))
    s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
    s.solve(pr.tolerance, 2000)
    s.set_profiling(1)
    acc = {}
    for _ in range(NSOLVES):
        st = s.solve(pr.tolerance, 2000)
        first = s.profile(first=True)
        for k, (n, ms) in s.profile().items():
            a = acc.setdefault(k, [0, 0.0]); a[0] += n - first[k][0]; a[1] += ms - first[k][1]
    info = s.get_info()
    if not acc:
        print("%-12s G %2d one solve, it %d" % (tag, G, info["iterations"]), flush=True)
        s.close()
        return
    it = sum(v[1] / v[0] for k, v in acc.items() if k != "probe" and v[0])
    print("%-12s G %2d status %d it %d | spmm_v4_dot %.4f spmm_v5_nrm_dot %.4f x_v6_v7 %.4f | iteration %.4f ms" % (
        tag, G, st, info["iterations"], acc["spmm_v4_dot"][1] / acc["spmm_v4_dot"][0], acc["spmm_v5_nrm_dot"][1] / acc["spmm_v5_nrm_dot"][0],
        acc["x_v6_v7"][1] / acc["x_v6_v7"][0], it), flush=True)
    s.close()


NSOLVES = int(sys.argv[3]) if len(sys.argv) > 3 else 3
Gs = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "4,8,12,16,24").split(",")]
want = (sys.argv[2] if len(sys.argv) > 2 else "natural,raster,strip4,strip8,strip16,tile4,tile8,morton").split(",")
ex = FDExample(16, 120, 4, 2, -0.25, 4)
pr = ex.problem()
x, y = ex.row_xyz[:, 0].astype(np.int64), ex.row_xyz[:, 1].astype(np.int64)
x0, y0 = x - x.min(), y - y.min()
keys = {"natural": np.arange(pr.mb), "raster": y0 * 1024 + x0, "morton": morton(x0, y0)}
for w in (4, 8, 16):
    keys["strip%d" % w] = ((x0 // w) * 1024 + y0) * w + x0 % w
for t in (4, 8):
    keys["tile%d" % t] = (((y0 // t) * 1024 + x0 // t) * t + y0 % t) * t + x0 % t
for name in want:
    ident = np.argsort(np.argsort(keys[name], kind="stable"), kind="stable").astype(np.int64)
    p = pr if name == "natural" else permuted(pr, ident)
    for G in Gs:
        run(name, p, G)
