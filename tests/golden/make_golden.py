#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REFERENCE (only possible in the build container, where
/root/reference exists and `make -C oracle ref` has produced oracle/_ref/).

What is stored is data only:
  *.xml        inputs written by the reference's own generator (example/tfqmrgpu_generate_FD_example.cxx)
  *.npz        outputs of the reference's compiled CPU library for those inputs (and for the synthetic
               inputs of tfqmrgpu_amd/problems.py): createPlan index lists, bufferSize, and full solves
               (status, iterations, residual, flops, solution blocks or their checksums).
Solves are run with the documented block semantics (A's flag flipped for the CPU library, see
oracle/pyoracle.py) and the reference CPU path's shadow vector (rand() from seed 1).
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import tfqmrgpu_amd as T  # noqa: E402  (only for the Problem container and the XML reader)
from tfqmrgpu_amd import problems as PR  # noqa: E402
from oracle import pyoracle as O  # noqa: E402

FD_FIXTURES = {  # name -> arguments of generate_FD_example: rsb rtb block_edge dim energy ref echo nFD
    "fd_8x8_3d":  (1.75, 6.75, 2, 3, 0.0, "n", 0, 4),     # the generator's defaults
    "fd_16x16_2d": (6, 24, 4, 2, -0.25, "n", 0, 4),
    "fd_16x16_small": (4, 12, 4, 2, -0.25, "n", 0, 4),     # 49 rows, 5 columns: quick checks
    "fd_4x4_2d": (3, 9, 2, 2, -0.1, "n", 0, 4),
}


def offset1(pr):
    return T.Problem(pr.rowPtrA + 1, pr.colIndA + 1, pr.A, pr.rowPtrX + 1, pr.colIndX + 1,
                     pr.rowPtrB + 1, pr.colIndB + 1, pr.B, None, pr.tolerance, 1)


def record(ref, name, pr, solves, keep_x=True):
    out = {}
    an = ref.analyse(pr)
    for k in ("pairs", "starts", "subset", "colindx", "original_bsrColIndX"):
        out["plan_" + k] = an[k]
    out["plan_nCols"] = np.int64(an["nCols"])
    an1 = ref.analyse(offset1(pr))
    for k in ("pairs", "starts", "subset", "colindx"):
        assert np.array_equal(an[k], an1[k])
    out["plan_original_bsrColIndX_off1"] = an1["original_bsrColIndX"]
    for prec, tol, maxit in solves:
        st, X, info = ref.solve_staged(pr, prec, threshold=tol, max_iterations=maxit)
        tag = "solve_%s_" % prec
        out[tag + "status"] = np.int64(st)
        out[tag + "threshold"] = np.float64(tol)
        out[tag + "maxit"] = np.int64(maxit)
        out[tag + "iterations"] = np.int64(info["iterations"])
        out[tag + "residual"] = np.float64(info["residual"])
        out[tag + "flops"] = np.float64(info["flops"])
        out[tag + "ref_buffer_bytes"] = np.int64(info["buffer_bytes"])
        out[tag + "sumX"] = np.complex128(X.sum())
        out[tag + "frobX"] = np.float64(np.sqrt((np.abs(X) ** 2).sum()))
        out[tag + "maxabsX"] = np.float64(np.abs(X).max())
        if keep_x:
            out[tag + "X"] = X.astype(np.complex128 if prec == "z" else np.complex64)
        else:
            out[tag + "X_sample"] = X.reshape(-1)[::97].copy()
        print("%-16s %s tol %.1e -> status %d, %d iterations, residual %.3e" % (name, prec, tol, st, info["iterations"], info["residual"]))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists")
    ref = O.Reference()
    tmp = "/tmp/tfq_golden"
    for name, args in FD_FIXTURES.items():
        xml = O.fd_xml(args, tmp)
        shutil.copy(xml, os.path.join(HERE, name + ".xml"))
        pr = PR.read_xml(xml)
        big = pr.nnzbX * pr.LM * pr.LN > 60000
        ctol = 1e-4 if pr.LM == 16 else 1e-2
        record(ref, name, pr, [("z", pr.tolerance, 2000), ("c", ctol, 200)], keep_x=not big)
    record(ref, "julia_kat", PR.julia_kat(), [("z", 1.2e-8, 210), ("c", 1.2e-5, 210)])
    record(ref, "dense_random", PR.dense_random(), [("z", 1e-10, 500), ("c", 1e-5, 500)])
    record(ref, "dense_random_rect", PR.dense_random(mb=4, LM=4, LN=8, ncols=3, seed=21), [("z", 1e-10, 500)])
    record(ref, "stencil_8x8", PR.stencil_2d(6, 5, 8, 8, 3, seed=3, radius=2.5), [("z", 1e-9, 300), ("c", 1e-4, 300)])
    record(ref, "stencil_8x32", PR.stencil_2d(5, 4, 8, 32, 2, seed=4), [("z", 1e-9, 300)])


if __name__ == "__main__":
    main()
