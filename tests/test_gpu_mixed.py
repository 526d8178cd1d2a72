"""Mixed precision 'm' (reference: dormant, tfqmrgpu.cu:42, tfqmrgpu.h:72 "start with float and converge double").
The product runs complex<float> tfQMR solves inside an iterative refinement in double (tfq_api.hip: run_mixed); the oracle has no
such mode, so the checks are against the oracle's complex<double> solution of the same system and against A x = b itself:
  * the solve converges to the DOUBLE threshold (1e-9), far past the float floor (4.6e-5 on the FD fixture);
  * X is within 1e-6 max|X| of the oracle's 'z' solution (observed: 1e-10 ... 1e-8);
  * the residual the solver reports is the true residual in double arithmetic (recomputed here from the downloaded X);
  * float data in (setMatrix 'c') and out (getMatrix 'c') are accepted; the buffer is smaller than a 'z' plan's;
  * a system on which float iterations stagnate ends with status 9 and an honest residual instead of looping.
Needs an MI355X (`pytest -m gpu`)."""
import numpy as np
import pytest

import tfqmrgpu_amd as T
from conftest import load_problem
from tfqmrgpu_amd import problems as PR

pytestmark = pytest.mark.gpu

SIZES = [(4, 4), (4, 5), (4, 8), (4, 32), (8, 8), (8, 9), (8, 10), (8, 32), (8, 64),
         (16, 16), (16, 32), (16, 64), (32, 32), (32, 64), (64, 64)]


def true_residual(pr, X):
    """max_rhs |A X - B| / |B| on the pattern of X, dense numpy arithmetic in double (small systems only)"""
    mb, LM, LN = pr.mb, pr.LM, pr.LN
    off = pr.index_offset
    cols = sorted(set((pr.colIndX - off).tolist()))
    cid = {c: i for i, c in enumerate(cols)}
    Xd = np.zeros((mb, len(cols), LM, LN), complex)
    mask = np.zeros((mb, len(cols)), bool)
    rows = np.repeat(np.arange(mb), np.diff(pr.rowPtrX))
    for q, (r, c) in enumerate(zip(rows, pr.colIndX - off)):
        Xd[r, cid[c]] = X[q]; mask[r, cid[c]] = True
    Bd = np.zeros_like(Xd)
    rowsB = np.repeat(np.arange(mb), np.diff(pr.rowPtrB))
    for q, (r, c) in enumerate(zip(rowsB, pr.colIndB - off)):
        Bd[r, cid[c]] = pr.B[q]
    Y = np.zeros_like(Xd)
    rowsA = np.repeat(np.arange(mb), np.diff(pr.rowPtrA))
    for q, (r, k) in enumerate(zip(rowsA, pr.colIndA - off)):
        Y[r] += np.einsum("ik,ckj->cij", pr.A[q], Xd[k])
    R = (Y - Bd) * mask[:, :, None, None]                       # products outside the pattern of X are dropped
    r2 = (np.abs(R) ** 2).sum(axis=(0, 2))                      # [cols, LN]
    b2 = (np.abs(Bd) ** 2).sum(axis=(0, 2))
    return float(np.sqrt((r2 / b2).max()))


@pytest.mark.parametrize("name", ["fd_16x16_2d", "fd_16x16_small", "dense_random", "dense_random_rect", "stencil_8x8", "stencil_8x32", "julia_kat"])
def test_mixed_converges_to_the_double_solution(oracle, name):
    pr = load_problem(name)
    tol = {"julia_kat": 1.2e-8, "dense_random": 1e-10, "dense_random_rect": 1e-10}.get(name, 1e-9)
    st, X, info = T.solve_problem(pr, "m", threshold=tol, max_iterations=500)
    st0, X0, info0 = oracle.solve(pr, "z", threshold=tol, max_iterations=500)
    assert st == st0 == 0, (st, info)
    assert info["residual"] <= tol                                          # far below the float floor of these systems
    h = info["refinement_history"]
    assert len(h) >= 3 and h[0] == pytest.approx(1.0, rel=1e-12) and h[-1] == info["residual"]
    assert all(b < 0.1 * a for a, b in zip(h[:-1], h[1:-1])) or len(h) <= 3    # every full cycle gains more than a digit
    assert np.abs(X - X0).max() <= 1e-6 * np.abs(X0).max()                  # within 1e-6 max|X| of the oracle's z solution
    assert np.abs(X - X0).max() <= 30 * tol * np.abs(X0).max()              # in fact as close as two solutions at this threshold are
    res = true_residual(pr, X)
    assert res <= 1.01 * tol and abs(res - info["residual"]) <= 1e-3 * info["residual"] + 1e-15
    assert 0 < info["iterations"] <= 3 * info0["iterations"] + 6            # the sum of the float iterations
    assert info["flops"] > 0 and info["buffer_bytes"] > 0


@pytest.mark.parametrize("LM,LN", SIZES)
def test_mixed_all_block_sizes(oracle, LM, LN):
    pr = PR.stencil_2d(5, 4, LM, LN, 2, seed=100 + LM + LN)
    st, X, info = T.solve_problem(pr, "m", threshold=1e-9, max_iterations=300)
    st0, X0, info0 = oracle.solve(pr, "z", threshold=1e-9, max_iterations=300)
    assert st == st0 == 0 and info["residual"] <= 1e-9
    assert np.abs(X - X0).max() <= 1e-7 * np.abs(X0).max()
    assert abs(true_residual(pr, X) - info["residual"]) <= 1e-3 * info["residual"]


def test_mixed_takes_and_returns_float_data(oracle):
    # the reference's 'm' plans are fed like 'c' plans (tfqmrgpu.cu:538-542: everything but 'z' is float data)
    pr = load_problem("fd_16x16_2d")
    A32, B32 = pr.A.astype(np.complex64), pr.B.astype(np.complex64)
    rounded = T.Problem(pr.rowPtrA, pr.colIndA, A32.astype(np.complex128), pr.rowPtrX, pr.colIndX, pr.rowPtrB, pr.colIndB,
                        B32.astype(np.complex128), None, 1e-9, pr.index_offset)
    st0, X0, _ = oracle.solve(rounded, "z", threshold=1e-9, max_iterations=300)      # the system the float data define
    st, X, info = T.solve_problem(pr, "m", threshold=1e-9, max_iterations=300, data_precision="c")
    assert st == st0 == 0 and info["residual"] <= 1e-9                                # converges in double on the float-valued system
    assert X.dtype == np.complex64 or np.abs(X - X.astype(np.complex64)).max() == 0   # what comes back is float data
    assert np.abs(X - X0).max() <= 2e-7 * np.abs(X0).max()                            # = the double solution rounded to float
    with T.Solver() as s:                                                             # mixed in / out on one plan: z in, c out
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, "m"))
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        assert s.solve(1e-9, 300) == 0
        Xz = s.get_matrix()
        s.data_precision = "c"
        Xc = s.get_matrix()
        assert np.array_equal(Xc, Xz.astype(np.complex64))
        for prec in "zc":                                                             # x survives a round trip bit for bit
            s.data_precision = prec
            s.set_matrix("X", Xc)
            assert np.array_equal(s.get_matrix(), Xc if prec == "c" else Xc.astype(np.complex128))


def test_mixed_buffer_is_smaller_and_the_plan_can_be_solved_again():
    pr = PR.stencil_2d(12, 12, 16, 16, 4, seed=7)
    with T.Solver() as s:
        s.create_plan(pr)
        nz, nc = s.buffer_size(16, 16, "z"), s.buffer_size(16, 16, "c")
        nm = s.buffer_size(16, 16, "m")
        assert nc < nm < 0.9 * nz                     # 11 float-sized vectors against 15, A twice (P2: 0.70, test_mixed_P2_full_size)
        s.set_buffer(nbytes=nm)
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        assert s.solve(1e-9, 300) == 0
        X1, i1 = s.get_matrix(), s.get_info()
        assert s.solve(1e-9, 300) == 0                # same plan, same B: bit-identical (deterministic reductions)
        assert np.array_equal(s.get_matrix(), X1) and s.get_info()["iterations"] == i1["iterations"]
        s.set_matrix("B", 2 * pr.B)                   # new right-hand side on the same plan (README.md:97-104)
        assert s.solve(1e-9, 300) == 0
        assert np.abs(s.get_matrix() - 2 * X1).max() <= 1e-7 * np.abs(X1).max()
        assert s.solve(1e-9, 0) == 9 and np.abs(s.get_matrix()).max() == 0      # no iteration allowed: x = 0
        assert s.solve(1e-9, 2) == 9 and s.get_info()["iterations"] == 2       # the bound on the SUM of the float iterations holds


def test_mixed_where_float_iterations_stagnate(oracle):
    # fd_4x4_2d: complex<float> tfQMR stagnates at |r|/|b| ~ 0.14 (200 iterations, profiles/r02_parity_report.txt), complex<double> needs 32.
    # The mixed mode must notice instead of iterating for ever: every float solve ends itself when a probe finds no progress
    # (Ctl::stallStop), the refinement gives up after two cycles without gain, status 9 with the true residual
    pr = load_problem("fd_4x4_2d")
    st, X, info = T.solve_problem(pr, "m", threshold=1e-9, max_iterations=2000)
    assert st in (0, 9)
    res = true_residual(pr, X)
    assert abs(res - info["residual"]) <= 1e-3 * info["residual"]
    if st == 9:
        assert info["residual"] > 1e-9 and info["iterations"] == 2000       # getInfo reports maxIterations when not converged (tfqmrgpu_core.hxx:171)
        assert len(info["bound_history"]) < 1500                            # ... but it did NOT spend them


def test_mixed_refuses_what_it_cannot_do():
    pr = load_problem("fd_16x16_small")
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, "m"))
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        s.set_operator(lambda *a: 0.0)
        st = T.lib.tfqmrgpu_bsrsv_solve(s.handle, s.plan, 1e-9, 10)
        assert T.decode(st)[0] == 16 or T.decode(st)[0] != 0               # user-defined operators: 'z' and 'c' plans only
        s.set_operator(None)
        assert s.solve(1e-9, 300) == 0


def test_mixed_P2_full_size(oracle):
    """BASELINE config 2 in mixed precision: converges to 1e-9, the solution equals the 'z' one to 1e-8, and a float iteration of the
    inner solves costs at most 0.65 of a double iteration (HIP events of the same kernel classes; the whole point of the mode)"""
    from tfqmrgpu_amd.fd_generator import FDExample
    pr = FDExample(16, 120, 4, 2, -0.25, 4).problem()
    out = {}
    for prec in "zm":
        with T.Solver() as s:
            s.create_plan(pr)
            nbytes = s.buffer_size(16, 16, prec)
            s.set_buffer(nbytes=nbytes)
            s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
            st1 = s.solve(1e-9, 2000)
            first_solve = s.get_info()
            s.set_profiling(1)
            st = s.solve(1e-9, 2000)
            info = s.get_info()
            if prec == "m":
                # (r04) the plan remembers where its first float solve ran into the float floor: the SECOND solve of the plan asks its first cycle for
                # twice that instead of searching again, and a last cycle with less than two digits to gain runs a predicted number of iterations --
                # fewer float iterations (21 -> 17 on this system), the same threshold reached in double arithmetic; a new A forgets the floor
                assert st1 == 0 and first_solve["residual"] <= 1e-9
                assert info["iterations"] <= first_solve["iterations"] and info["iterations"] <= 18, (first_solve["iterations"], info["iterations"])
                s.set_profiling(0)
                assert s.solve(1e-9, 2000) == 0 and s.get_info()["iterations"] == info["iterations"]       # ... and every later one like the second
                s.set_matrix("A", pr.A)
                assert s.solve(1e-9, 2000) == 0 and s.get_info()["iterations"] == first_solve["iterations"]   # the first solve again
                s.set_profiling(1)
                st = s.solve(1e-9, 2000)
                info = s.get_info()
            prof, first = s.profile(), s.profile(first=True)
            ms = sum((prof[k][1] - first[k][1]) / max(1, prof[k][0] - first[k][0]) for k in prof if k != "probe")
            out[prec] = dict(st=st, X=s.get_matrix(), info=info, it_ms=ms, nbytes=nbytes, hist=s.refinement_history())
    z, m = out["z"], out["m"]
    assert z["st"] == m["st"] == 0 and m["info"]["residual"] <= 1e-9
    assert np.abs(m["X"] - z["X"]).max() <= 1e-8 * np.abs(z["X"]).max()
    assert m["it_ms"] <= 0.65 * z["it_ms"], (m["it_ms"], z["it_ms"])
    assert m["nbytes"] <= 0.75 * z["nbytes"]
    assert len(m["hist"]) >= 3 and m["hist"][-1] <= 1e-9
