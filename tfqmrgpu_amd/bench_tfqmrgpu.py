#!/usr/bin/env python3
"""Counterpart of the reference's benchmark driver (tfQMRgpu/source/bench_tfqmrgpu.cu:442-590), same positional
arguments and result lines, running on this library through its C-ABI:

  python -m tfqmrgpu_amd.bench_tfqmrgpu multi <planfile[.gz]> [precision=f] [nrep=1] [nsamp=1] [lm=16] [ln=lm]
      times the block-sparse multiply on a plan file `#nnzb_for_Y_A_X= nY nA nX` + lines `iY iA iX beta`
      (bench_tfqmrgpu.cu:456-498), cos/sin fill (:274-287), host re-computation check maxdev <= 1e-4 (:349-420)
  python -m tfqmrgpu_amd.bench_tfqmrgpu tfQMR <problem.xml> [precision=z] [nrep=1] [MaxIter=2000]
      solves the <LinearProblem> through the staged API with trans 'n' (row-major XML blocks, see DESIGN.md on the
      reference's 't'), compares with the stored X if the file has one (:178-205)
"""
import gzip
import sys
import time

import numpy as np


def _read_plan(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        head = f.readline().split()
        rows = np.loadtxt(f, dtype=np.int64)
    nY, nA, nX = int(head[1]), int(head[2]), int(head[3])
    change = np.concatenate([[True], rows[1:, 0] != rows[:-1, 0]])
    assert np.all(rows[change, 3] == 0) and np.all(rows[~change, 3] == 1), "beta must be 0 on the first line of a group"
    starts = np.concatenate([np.nonzero(change)[0], [len(rows)]]).astype(np.uint32)
    assert len(starts) == nY + 1
    return nY, nA, nX, starts, np.ascontiguousarray(rows[:, 1:3].reshape(-1).astype(np.uint32))


def multi(argv):
    import torch
    import tfqmrgpu_amd as T
    path = argv[2] if len(argv) > 2 else "plan"
    fF = (argv[3] if len(argv) > 3 else "f")[0].lower()
    nrep = int(argv[4]) if len(argv) > 4 else 1
    nsamp = int(argv[5]) if len(argv) > 5 else 1
    lm = int(argv[6]) if len(argv) > 6 else 16
    ln = int(argv[7]) if len(argv) > 7 else lm
    prec = "z" if fF in "dz" else "c"
    nY, nA, nX, starts, pairs = _read_plan(path)
    nPairs = len(pairs) // 2
    real = np.float64 if prec == "z" else np.float32
    print("\n# bench_multi<%d,%d> on GPU !!!!" % (lm, ln))
    print("# Execute %d repetitions, sample %d times." % (nrep, nsamp))

    def fill(n, rows, cols):
        arg = np.arange(n * rows * cols, dtype=np.float64).reshape(n, rows, cols)
        return np.stack([np.cos(arg), np.sin(arg)], axis=1).astype(real)
    A, X = fill(nA, lm, lm), fill(nX, lm, ln)
    dA, dX = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
    dY = torch.zeros((nY, 2, lm, ln), dtype=dA.dtype, device="cuda")
    dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
    s = T.Solver()
    times, nflop = [], 0.0
    # the launch order is prepared once, in front of the timed loop, as the reference prepares its launch there (bench_tfqmrgpu.cu:442-556 in front of
    # :289-440); mode 4: the library chooses how the XCDs split the listing; BENCH_ORDER=0: the caller's order.  Same bits either way.
    import ctypes as C
    import os
    order = C.c_void_p(None)
    T._check(T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, prec.encode(), lm, ln, nY, dS.data_ptr(), dP.data_ptr(), int(os.environ.get("BENCH_ORDER", 4)), C.byref(order)), "multiplyPrepare")
    print("# launch order: %s" % ("prepared by the library (tfqmrgpuExt_multiplyPrepare)" if order.value else "the listing's own"))
    fn = T.lib.tfqmrgpuExt_multiplyOrdered      # arguments bound once: the loop below should time the library, not Python
    call = (s.handle, prec.encode(), lm, ln, nY, dS.data_ptr(), dP.data_ptr(), dA.data_ptr(), dX.data_ptr(), dY.data_ptr(), order)
    T._check(fn(*call), "multiply")      # warm-up (module load), as the reference's first sample is
    for _ in range(nsamp):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        status = 0
        for _ in range(nrep):
            status |= fn(*call)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        T._check(status, "multiply")
        nflop += nrep * nPairs * 8.0 * lm * lm * ln
    tsum, tavg = sum(times), sum(times) / nsamp
    print("# GPU needed %.3f seconds, %.6f +/- %.6f sec per sample" % (tsum, tavg, float(np.std(times))))
    # host re-computation: Y[iY] = sum_p A[iA]^T-stored . X[iX]   (matA is stored transposed, :380-382)
    Ac = (A[:, 0] + 1j * A[:, 1]).transpose(0, 2, 1).astype(np.complex128)
    Xc = (X[:, 0] + 1j * X[:, 1]).astype(np.complex128)
    prod = np.einsum("pik,pkj->pij", Ac[pairs[0::2]], Xc[pairs[1::2]])
    Yr = np.zeros((nY, lm, ln), np.complex128)
    np.add.at(Yr, np.repeat(np.arange(nY), np.diff(starts.astype(np.int64))), prod)
    Yg = dY.cpu().numpy()
    maxdev = max(np.abs(Yg[:, 0] - Yr.real).max(), np.abs(Yg[:, 1] - Yr.imag).max())
    print("# GPU maxdev %g" % maxdev)
    if maxdev > 1e-4:
        print("# Warning! GPU result has large deviations (%g) for blockDim=%d x %d" % (maxdev, lm, ln))
        return 1
    ch = "F" if prec == "z" else "f"
    print("# GPU performed %.3f T%clop in %.3f seconds" % (nflop * 1e-12, ch, tsum))
    print("# GPU performance (lm,ln,tune)=(%3d,%3d,%d) is  %.1f G%clop/sec" % (lm, ln, 0, nflop * 1e-9 / tsum, ch))
    T.lib.tfqmrgpuExt_multiplyRelease(order)
    s.close()
    return 0


def tfqmr(argv):
    import tfqmrgpu_amd as T
    from tfqmrgpu_amd import problems as PR
    path = argv[2] if len(argv) > 2 else "problem"
    prec = (argv[3] if len(argv) > 3 else "z")[0].lower()
    prec = "z" if prec in "dz" else "m" if prec == "m" else "c"   # m: mixed precision, float arrays in and out like the reference's driver
    maxiter = int(argv[5]) if len(argv) > 5 else 2000
    print("\n# read file '%s' as input." % path)
    pr = PR.read_xml(path)
    print("# found tolerance= %g" % pr.tolerance)
    print("# requested precision= '%c' for LM= %d, LN= %d" % (prec, pr.LM, pr.LN))
    print("\n# nnzb for A=%d, X=%d, B=%d" % (pr.nnzbA, pr.nnzbX, pr.nnzbB))
    with T.Solver() as s:
        s.create_plan(pr)
        nbytes = s.buffer_size(pr.LM, pr.LN, prec)
        if prec == "m":
            s.data_precision = "c"
        print("# use %.6f GByte GPU memory" % (nbytes * 1e-9))
        s.set_buffer(nbytes=nbytes)
        s.set_matrix("A", pr.A, "n")
        s.set_matrix("B", pr.B, "n")
        t0 = time.perf_counter()
        st = s.solve(pr.tolerance, maxiter)
        dt = time.perf_counter() - t0
        T.lib.tfqmrgpuPrintError(st)
        X = s.get_matrix()
        info = s.get_info()
    if pr.X is not None and np.abs(pr.X).max() > 0:
        dev = np.abs(X - pr.X)
        print("# GPU maxdev %g avgdev %g" % (dev.max(), dev.mean()))
        if dev.max() >= 1e-5:
            return 1
    print("# GPU converged to %.1e in %d iterations" % (info["residual"], info["iterations"]))
    ch = "F" if prec == "z" else "f"
    print("# GPU performed %.3f T%clop in %.3f seconds = %.3f T%clop/s" % (info["flops"] * 1e-12, ch, dt, info["flops"] * 1e-12 / max(dt, 1e-6), ch))
    return st


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    if len(argv) < 2:
        print("Usage:  %s  [tfQMR/multiply]  [file]  [float/double]  [#repetitions]  [#iterations]  [#blocksize]" % argv[0])
        return 1
    return multi(argv) if argv[1][0] == "m" else tfqmr(argv)


if __name__ == "__main__":
    sys.exit(main())
