"""Parity of the HIP path (through the C-ABI of libtfQMRgpu.so) with the CPU oracle and the golden
vectors of the reference.  Needs an MI355X: run with `pytest -m gpu`.

Tolerances: integer results bit-exact; complex<double> solutions within
1e-7*max|X| at threshold 1e-9, complex<float> within 1e-3*max|X|; reported residual within 1e-6
relative and equal iteration counts when both sides use the same shadow vector (glibc mode)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

import tfqmrgpu_amd as T
from conftest import ALL_NAMES, GOLDEN, golden_solves, load_golden, load_problem
from tfqmrgpu_amd import problems as PR

pytestmark = pytest.mark.gpu

SIZES = [(4, 4), (4, 5), (4, 8), (4, 32), (8, 8), (8, 9), (8, 10), (8, 32), (8, 64),
         (16, 16), (16, 32), (16, 64), (32, 32), (32, 64), (64, 64)]


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a GPU; there is no CPU fallback"
    torch.cuda.set_device(0)
    return torch


def _tol(prec):
    return 1e-7 if prec == "z" else 1e-3


from tolerances import C_TOL, Z_TOL  # noqa: E402  (shared with tests/test_oracle_sensitivity.py, which runs without a GPU)


@pytest.mark.parametrize("name", ALL_NAMES)
def test_solve_matches_oracle_and_golden(oracle, name):
    pr, g = load_problem(name), load_golden(name)
    for prec, tol, maxit in golden_solves(g):
        tag = "solve_%s_" % prec
        st, X, info = T.solve_problem(pr, prec, threshold=tol, max_iterations=maxit, shadow_mode=T.SHADOW_GLIBC_RAND)
        st0, X0, info0 = oracle.solve(pr, prec, threshold=tol, max_iterations=maxit)
        assert st == st0 == int(g[tag + "status"])
        if st != 0:
            continue  # a run into maxIterations (float floor) has no meaningful solution to compare
        scale = float(g[tag + "maxabsX"])
        xtol = 1e-7 if prec == "z" else C_TOL[name]["x"]
        assert np.abs(X - X0).max() <= xtol * scale, (name, prec)
        if tag + "X" in g:
            assert np.abs(X - g[tag + "X"]).max() <= xtol * scale
        else:
            assert np.abs(X.reshape(-1)[::97] - g[tag + "X_sample"]).max() <= xtol * scale
        assert info["residual"] <= tol
        h, h0 = info["bound_history"], info0["bound_history"]
        if prec == "z":
            t = Z_TOL[name]
            assert info["iterations"] == info0["iterations"] == int(g[tag + "iterations"])
            assert info["flops"] == float(g[tag + "flops"])
            assert len(h) == len(h0)
            half = (len(h) + 1) // 2
            assert np.allclose(h[:half], h0[:half], rtol=t["half"], atol=0), (name, np.abs(h[:half] / h0[:half] - 1).max())
            assert np.allclose(h, h0, rtol=t["hist"], atol=0), (name, np.abs(h / h0 - 1).max())
            # (not pytest.approx: its default absolute tolerance of 1e-12 would swallow residuals of 1e-10)
            assert abs(info["residual"] - info0["residual"]) <= t["res"] * info0["residual"], (name, info["residual"] / info0["residual"] - 1)
            assert abs(info["residual"] - float(g[tag + "residual"])) <= t["res"] * float(g[tag + "residual"]), name
        else:
            d = abs(info["iterations"] - info0["iterations"])
            assert d <= C_TOL[name]["it"], "%s: %d against %d iterations in complex<float>: the float trajectories separate (see C_TOL)" % (
                name, info["iterations"], info0["iterations"])
            assert np.allclose(h[:2], h0[:2], rtol=1e-3, atol=0)


@pytest.mark.parametrize("name", ["fd_16x16_small", "fd_8x8_3d", "dense_random", "julia_kat"])
def test_default_shadow_vector_converges_to_the_same_solution(oracle, name):
    pr = load_problem(name)
    tol = {"julia_kat": 1.2e-8, "dense_random": 1e-10}.get(name, pr.tolerance)
    st, X, info = T.solve_problem(pr, "z", threshold=tol, max_iterations=500)
    st0, X0, info0 = oracle.solve(pr, "z", threshold=tol, max_iterations=500)
    assert st == st0 == 0
    assert abs(info["iterations"] - info0["iterations"]) <= 3
    assert info["residual"] <= tol
    assert np.abs(X - X0).max() <= 1e-6 * np.abs(X0).max()


def test_known_answers():
    pr = PR.julia_kat()  # example/tfqmrgpu_Julia_example.jl:117-120
    st, X, info = T.solve_problem(pr, "z", threshold=1.2e-8, max_iterations=210)
    assert st == 0
    line = np.arange(1, 8) / 8.0
    for j in range(5):
        assert np.abs(X[:, j % 4, j] - line * 1j ** (j // 4)).max() < 1e-9
    st, X, info = T.solve_problem(pr, "c", threshold=1.2e-5, max_iterations=210)
    assert st == 0 and np.abs(X[:, 0, 0] - line).max() < 1e-4
    pr = PR.dense_random()  # example/tfqmrgpu_Fortran_example.F90:108-126: A*X == B, A not symmetric
    Xd = PR.dense_reference_solution(pr)
    st, X, info = T.solve_problem(pr, "z", threshold=1e-10, max_iterations=500)
    assert st == 0 and np.abs(X - Xd).max() < 1e-8 * np.abs(Xd).max()


@pytest.mark.parametrize("trans", ["n", "t", "c", "h", "*", "N", "T"])
def test_trans_flags_of_A(trans):
    pr = PR.dense_random(mb=4, LM=8, LN=8, ncols=2, seed=31)
    op = {"n": lambda a: a, "t": lambda a: a.transpose(0, 2, 1), "c": lambda a: a.conj().transpose(0, 2, 1),
          "h": lambda a: a.conj().transpose(0, 2, 1), "*": lambda a: a.conj()}[trans.lower()]
    want = PR.dense_reference_solution(T.Problem(pr.rowPtrA, pr.colIndA, op(pr.A), pr.rowPtrX, pr.colIndX,
                                                 pr.rowPtrB, pr.colIndB, pr.B, None, 1e-10))
    st, X, info = T.solve_problem(pr, "z", threshold=1e-10, max_iterations=500, transA=trans)
    assert st == 0 and np.abs(X - want).max() < 1e-8 * np.abs(want).max()


def _user_array(blocks, layout, trans):
    """what a caller would hand to setMatrix for complex blocks [n, R, C]"""
    m = {"n": blocks, "t": blocks.transpose(0, 2, 1), "*": blocks.conj(), "c": blocks.conj().transpose(0, 2, 1)}[trans]
    if layout == T.LAYOUT_RIRIRIRI:
        return np.stack([m.real, m.imag], axis=-1).reshape(len(m), -1)
    if layout == T.LAYOUT_RRRRIIII:
        return np.stack([m.real, m.imag], axis=1).reshape(len(m), -1)
    return np.stack([m.real, m.imag], axis=2).reshape(len(m), -1)  # RRIIRRII


@pytest.mark.parametrize("prec", ["z", "c"])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (8, 10), (8, 32), (8, 64), (16, 16), (16, 32), (32, 32), (16, 64), (32, 64), (64, 64)])   # 16 x 16, 8 x 8 | 32 | 64 z / 16 x 16, 16 | 32 | 64 x 32 | 64 c: plans with groups of rows interleaved
def test_set_get_matrix_layouts(prec, shape):
    LM, LN = shape
    pr = PR.stencil_2d(4, 3, LM, LN, 3, seed=12, radius=1.6)
    rng = np.random.default_rng(3)
    Xv = rng.standard_normal((pr.nnzbX, LM, LN)) + 1j * rng.standard_normal((pr.nnzbX, LM, LN))
    real = np.float64 if prec == "z" else np.float32
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(LM, LN, prec))
        for lay_in in (T.LAYOUT_RIRIRIRI, T.LAYOUT_RRRRIIII, T.LAYOUT_RRIIRRII):
            for tr_in in "nt*c":
                s.set_matrix("X", _user_array(Xv, lay_in, tr_in).astype(real), tr_in, lay_in)
                for lay_out in (T.LAYOUT_RIRIRIRI, T.LAYOUT_RRRRIIII, T.LAYOUT_RRIIRRII):
                    for tr_out in "nt*c":
                        got = s.get_matrix(trans=tr_out, layout=lay_out, raw=True)
                        want = _user_array(Xv, lay_out, tr_out).astype(real)
                        assert np.array_equal(got, want), (lay_in, tr_in, lay_out, tr_out)


@pytest.mark.parametrize("prec", ["z", "c", "m"])
def test_set_get_matrix_take_device_arrays(torch_cuda, prec):
    """setMatrix / getMatrix with arrays in DEVICE memory (include/tfqmrgpu.h): converted in place of the caller's array on the stream,
    same result as with host arrays -- the "same A, new B, solve again" loop without a PCIe crossing"""
    torch = torch_cuda
    pr = PR.stencil_2d(6, 5, 16, 16, 3, seed=21)
    ctype = torch.complex64 if prec == "c" else torch.complex128
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, prec))
        s.set_matrix("A", pr.A); s.set_matrix("B", pr.B)
        tol = 1e-4 if prec == "c" else 1e-9
        assert s.solve(tol, 200) == 0
        X_host = s.get_matrix()
        dA = torch.from_numpy(pr.A).to(ctype).cuda()
        dB = torch.from_numpy(pr.B).to(ctype).cuda()
        dX = torch.zeros((pr.nnzbX, 16, 16), dtype=ctype, device="cuda")
        s.set_matrix_device("A", dA.data_ptr())
        s.set_matrix_device("B", dB.data_ptr())
        assert s.solve(tol, 200) == 0
        s.get_matrix_device(dX.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(dX.cpu().numpy(), X_host)
        for trans in "tc":                                     # the layout / transposition machinery is the same kernel
            s.get_matrix_device(dX.data_ptr(), trans=trans)
            torch.cuda.synchronize()
            assert np.array_equal(dX.cpu().numpy().reshape(pr.nnzbX, -1), s.get_matrix(trans=trans).reshape(pr.nnzbX, -1))
        s.set_matrix_device("B", (2 * dB).data_ptr())           # new right-hand side, resident on the device
        assert s.solve(tol, 200) == 0
        s.get_matrix_device(dX.data_ptr())
        torch.cuda.synchronize()
        assert np.abs(dX.cpu().numpy() - 2 * X_host).max() <= (1e-3 if prec == "c" else 1e-7) * np.abs(X_host).max()


def _rand_pairs(rng, nY, nA, nX, maxp):
    starts, pairs = [0], []
    for _ in range(nY):
        k = int(rng.integers(0, maxp + 1))
        for _ in range(k):
            pairs += [int(rng.integers(0, nA)), int(rng.integers(0, nX))]
        starts.append(len(pairs) // 2)
    return np.array(starts, np.uint32), np.array(pairs + [0, 0], np.uint32)[: max(2, len(pairs))]


@pytest.mark.parametrize("prec", ["z", "c"])
@pytest.mark.parametrize("size", SIZES)
def test_multiply_all_block_sizes(torch_cuda, oracle, prec, size):
    torch = torch_cuda
    LM, LN = size
    rng = np.random.default_rng(LM * 100 + LN)
    nY, nA, nX = 37, 23, 29
    starts, pairs = _rand_pairs(rng, nY, nA, nX, 6)
    real = np.float64 if prec == "z" else np.float32
    A = rng.uniform(-1, 1, (nA, 2, LM, LM)).astype(real)
    X = rng.uniform(-1, 1, (nX, 2, LM, LN)).astype(real)
    want = oracle.spmm(prec, LM, LN, starts, pairs, A, X.copy())[:nY] if nY <= nX else None
    Yw = np.zeros((nY, 2, LM, LN), real)
    Yw[:] = oracle.spmm(prec, LM, LN, starts, pairs, A, np.concatenate([X, np.zeros((max(0, nY - nX), 2, LM, LN), real)]))[:nY]
    dA, dX = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
    dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
    dY = torch.full((nY, 2, LM, LN), 7.0, dtype=dA.dtype, device="cuda")
    with T.Solver() as s:
        st = T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(),
                                        dA.data_ptr(), dX.data_ptr(), dY.data_ptr())
        assert st == 0
        torch.cuda.synchronize()
        got = dY.cpu().numpy()
        # ... and with a prepared launch order (tfqmrgpuExt_multiplyPrepare, every mode): which work group computes a Y block changes, no bit of it does
        # (shapes whose kernel takes no order get a null order = the caller's)
        for mode in (1, 2, 3, 4):
            order = C.c_void_p(None)
            assert T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(), mode, C.byref(order)) == 0
            assert bool(order.value) == (LM % 16 == 0 and LN % 16 == 0)
            dZ = torch.full((nY, 2, LM, LN), 5.0, dtype=dA.dtype, device="cuda")
            assert T.lib.tfqmrgpuExt_multiplyOrdered(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(),
                                                     dA.data_ptr(), dX.data_ptr(), dZ.data_ptr(), order) == 0
            torch.cuda.synchronize()
            assert np.array_equal(dZ.cpu().numpy(), got), (size, prec, mode)
            assert T.lib.tfqmrgpuExt_multiplyRelease(order) == 0
    eps = 1e-13 if prec == "z" else 2e-5
    assert np.abs(got - Yw).max() <= eps * LM * 6, (size, prec)
    del want


def _load_plan_file():
    # test/multiplication/plan_unordered.14-287-16 of the reference: "#nnzb_for_Y_A_X= nY nA nX", then iY iA iX beta
    with gzip.open(os.path.join(GOLDEN, "plan_unordered.14-287-16.gz"), "rt") as f:
        head = f.readline().split()
        rows = np.loadtxt(f, dtype=np.int64)
    nY, nA, nX = int(head[1]), int(head[2]), int(head[3])
    starts, prev = [], -1
    for n, (iy, ia, ix, beta) in enumerate(rows):  # groups by change of iY (bench_tfqmrgpu.cu:480-495)
        if iy != prev:
            assert beta == 0
            starts.append(n)
            prev = iy
        else:
            assert beta == 1
    starts.append(len(rows))
    return nY, nA, nX, np.array(starts, np.uint32), np.ascontiguousarray(rows[:, 1:3].reshape(-1).astype(np.uint32))


@pytest.mark.parametrize("prec", ["z", "c"])
def test_multiply_reference_plan_file(torch_cuda, oracle, prec):
    # BASELINE config 1: the reference's multiplication benchmark input, cos/sin fill (bench_tfqmrgpu.cu:274-287),
    # its acceptance test is maxdev <= 1e-4 (bench_tfqmrgpu.cu:414)
    torch = torch_cuda
    nY, nA, nX, starts, pairs = _load_plan_file()
    assert (nY, nA, nX, len(pairs) // 2) == (4490, 13109, 4490, 50526)
    LM = LN = 16
    real = np.float64 if prec == "z" else np.float32

    def fill(n):
        arg = np.arange(n * LM * LN, dtype=np.float64).reshape(n, LM, LN)
        return np.stack([np.cos(arg), np.sin(arg)], axis=1).astype(real)
    A, X = fill(nA), fill(nX)
    want = oracle.spmm(prec, LM, LN, starts, pairs, A, X)
    dA, dX = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
    dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
    dY = torch.zeros_like(dX)
    with T.Solver() as s:
        assert T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(),
                                          dA.data_ptr(), dX.data_ptr(), dY.data_ptr()) == 0
        torch.cuda.synchronize()
        got = dY.cpu().numpy()
        # with a prepared order (mode 4: the library chooses how the XCDs split the listing -- here the rows, A outweighs X): the same bits; an order
        # is refused for another listing
        for mode in (1, 3, 4):
            order = C.c_void_p(None)
            assert T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(), mode, C.byref(order)) == 0 and order.value
            dZ = torch.zeros_like(dX)
            assert T.lib.tfqmrgpuExt_multiplyOrdered(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(),
                                                     dA.data_ptr(), dX.data_ptr(), dZ.data_ptr(), order) == 0
            torch.cuda.synchronize()
            assert np.array_equal(dZ.cpu().numpy(), got), mode
            assert T.decode(T.lib.tfqmrgpuExt_multiplyOrdered(s.handle, prec.encode(), LM, LN, nY - 1, dS.data_ptr(), dP.data_ptr(),
                                                              dA.data_ptr(), dX.data_ptr(), dZ.data_ptr(), order))[0] == 7
            assert T.lib.tfqmrgpuExt_multiplyRelease(order) == 0
    dev = np.abs(got - want).max()
    assert dev <= (1e-12 if prec == "z" else 1e-4), dev


@pytest.mark.parametrize("prec", ["z", "c"])
@pytest.mark.parametrize("size", SIZES)
def test_solve_all_block_sizes(oracle, prec, size):
    LM, LN = size
    pr = PR.stencil_2d(4, 4, LM, LN, 2, seed=LM + LN, radius=2.3)
    tol = 1e-9 if prec == "z" else 1e-4
    st, X, info = T.solve_problem(pr, prec, threshold=tol, max_iterations=100, shadow_mode=T.SHADOW_GLIBC_RAND)
    st0, X0, info0 = oracle.solve(pr, prec, threshold=tol, max_iterations=100)
    assert st == st0 == 0, (st, st0)
    assert abs(info["iterations"] - info0["iterations"]) <= (0 if prec == "z" else 1)
    assert np.abs(X - X0).max() <= _tol(prec) * np.abs(X0).max()


@pytest.mark.parametrize("prec", ["z", "c"])
def test_one_call_driver(prec):
    # tfqmrgpu_bsrsv_z / _c the way the reference's Julia and Python examples call them
    # (example/tfqmrgpu_Julia_example.jl:88-120, example/tfqmrgpu_python_example.py:40-67)
    pr = PR.julia_kat()
    real, cplx = (np.float64, np.complex128) if prec == "z" else (np.float32, np.complex64)
    A = np.ascontiguousarray(pr.A.astype(cplx))
    B = np.ascontiguousarray(pr.B.astype(cplx))
    X = np.zeros((pr.nnzbX, pr.LM, pr.LN), dtype=cplx)
    it = C.c_int32(210)
    res = C.c_float(1.2e-8 if prec == "z" else 1.2e-5)
    fn = T.lib.tfqmrgpu_bsrsv_z if prec == "z" else T.lib.tfqmrgpu_bsrsv_c
    st = fn(pr.mb, pr.LM, pr.LN, T._ptr(pr.rowPtrA), pr.nnzbA, T._ptr(pr.colIndA), T._ptr(A), b"n",
            T._ptr(pr.rowPtrX), pr.nnzbX, T._ptr(pr.colIndX), T._ptr(X), b"n",
            T._ptr(pr.rowPtrB), pr.nnzbB, T._ptr(pr.colIndB), T._ptr(B), b"n", C.byref(it), C.byref(res), 0, 0)
    assert st == 0 and 0 < it.value <= 12
    assert res.value <= (1.2e-8 if prec == "z" else 1.2e-5)
    line = np.arange(1, 8) / 8.0
    assert np.abs(X[:, 0, 0] - line).max() < (1e-9 if prec == "z" else 1e-4)
    assert np.abs(X[:, 0, 4] - 1j * line).max() < (1e-9 if prec == "z" else 1e-4)   # fifth RHS carries the phase i
    # Fortran indices through the same entry point (iterations / residual are in-out arguments: reset them)
    it = C.c_int32(210)
    res = C.c_float(1.2e-8 if prec == "z" else 1.2e-5)
    st = fn(pr.mb, pr.LM, pr.LN, T._ptr(pr.rowPtrA + 1), pr.nnzbA, T._ptr(pr.colIndA + 1), T._ptr(A), b"n",
            T._ptr(pr.rowPtrX + 1), pr.nnzbX, T._ptr(pr.colIndX + 1), T._ptr(X), b"n",
            T._ptr(pr.rowPtrB + 1), pr.nnzbB, T._ptr(pr.colIndB + 1), T._ptr(B), b"n", C.byref(it), C.byref(res), 1, 0)
    assert st == 0 and np.abs(X[:, 0, 0] - line).max() < (1e-9 if prec == "z" else 1e-4)
    # an unsupported block shape comes back as 12 with LM / LN in the word (tfqmrgpu.cu:70)
    it = C.c_int32(10)
    st = fn(pr.mb, 3, 3, T._ptr(pr.rowPtrA), pr.nnzbA, T._ptr(pr.colIndA), T._ptr(A), b"n",
            T._ptr(pr.rowPtrX), pr.nnzbX, T._ptr(pr.colIndX), T._ptr(X), b"n",
            T._ptr(pr.rowPtrB), pr.nnzbB, T._ptr(pr.colIndB), T._ptr(B), b"n", C.byref(it), C.byref(res), 0, 0)
    assert T.decode(st) == (12, 3, 3)


def test_profiling_levels_change_no_result():
    """tfqmrgpuExt_setProfiling: 1 = HIP events around every kernel class, 2 = only around the two fused multiplies (what
    bench.py keeps inside its timed region); the events never change what is computed"""
    pr = load_problem("fd_16x16_2d")
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, "z"))
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        got = {}
        for level in (0, 1, 2):
            s.set_profiling(level)
            assert s.solve(1e-9, 2000) == 0
            got[level] = (s.get_info()["iterations"], s.get_matrix(), s.profile(), s.profile(gated=True))
        it = got[0][0]
        assert got[1][0] == got[2][0] == it and np.array_equal(got[0][1], got[1][1]) and np.array_equal(got[0][1], got[2][1])
        assert all(n == 0 for n, _ in got[0][2].values())
        for k, (n, ms) in got[1][2].items():
            assert (n == it or k == "probe") and ms > 0, k            # every class once per iteration, probes when requested
        for k, (n, ms) in got[2][2].items():
            assert (n, ms > 0) == ((it, True) if k in ("spmm_v4_dot", "spmm_v5_nrm_dot") else (0, False)), k
        # launches enqueued ahead of the stopping decision return at once: counted apart, same number in both levels
        assert got[1][3]["spmm_v4_dot"][0] == got[2][3]["spmm_v4_dot"][0]


def test_plan_reuse_and_status_codes():
    pr = load_problem("fd_16x16_small")
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, "z"))
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        assert s.solve(1e-9, 2000) == 0
        first, X1 = s.get_info(), s.get_matrix()
        assert s.solve(1e-9, 2000) == 0                      # a second solve on the same plan (README.md:97-104)
        again, X2 = s.get_info(), s.get_matrix()
        assert first["iterations"] == again["iterations"] and np.array_equal(X1, X2)
        assert again["flops_all"] == 2 * again["flops"]
        s.set_matrix("B", 2 * pr.B)                          # new right-hand sides, same A
        assert s.solve(1e-9, 2000) == 0
        assert np.abs(s.get_matrix() - 2 * X1).max() <= 1e-7 * np.abs(X1).max()
        assert s.solve(1e-9, 3) == 9                         # TFQMRGPU_STATUS_MAX_ITERATIONS, bare (tfqmrgpu_core.hxx:170)
        assert s.get_info()["iterations"] == 3
        assert s.solve(1e-9, 0) == 9
        # precision mismatch: code 16 with the offending character (tfqmrgpu.cu:538-542)
        a = np.zeros((pr.nnzbA, 16, 16, 2), np.float32)
        st = T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"A", T._ptr(a), b"c", 16, 16, b"n", 0x55)
        assert T.decode(st)[::2] == (16, ord("c"))
        s.set_matrix("A", 0 * pr.A)                          # A == 0: v3.(A v6) == 0 for every RHS -> breakdown
        assert s.solve(1e-9, 50) == 6                        # TFQMRGPU_STATUS_BREAKDOWN (tfqmrgpu_core.hxx:258)


def test_reduce_paths_single_rank(torch_cuda):
    # the multi-GPU stopping test with one rank: host callback and RCCL must not change anything
    pr = load_problem("fd_16x16_small")
    st0, X0, info0 = T.solve_problem(pr, "z")
    calls = []

    def cb(ctx, values, n):
        calls.append([values[i] for i in range(n)])
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, "z"))
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        keep = T.REDUCE_CB(cb)
        assert T.lib.tfqmrgpuExt_setReduceCallback(s.handle, keep, None) == 0
        assert s.solve(pr.tolerance, 2000) == 0
        assert s.get_info()["iterations"] == info0["iterations"] and np.array_equal(s.get_matrix(), X0)
        assert len(calls) >= 2 * info0["iterations"]
        assert T.lib.tfqmrgpuExt_setReduceCallback(s.handle, C.cast(None, T.REDUCE_CB), None) == 0
        uid = (C.c_char * 128)()
        assert T.lib.tfqmrgpuExt_commUniqueId(uid) == 0
        assert T.lib.tfqmrgpuExt_commInit(s.handle, 1, 0, uid) == 0
        assert s.solve(pr.tolerance, 2000) == 0
        assert s.get_info()["iterations"] == info0["iterations"] and np.array_equal(s.get_matrix(), X0)
        assert T.lib.tfqmrgpuExt_commDestroy(s.handle) == 0


def test_large_problem_properties(torch_cuda, oracle):
    # size-independent checks at a size the oracle cannot solve in seconds: the solution must satisfy
    # A*X == B on the pattern (checked with the device multiply, itself checked above), and the
    # multiply must be linear
    torch = torch_cuda
    pr = PR.stencil_2d(48, 48, 16, 16, 24, seed=8, radius=9.0)
    st, X, info = T.solve_problem(pr, "z", threshold=1e-9, max_iterations=300)
    assert st == 0 and info["residual"] <= 1e-9
    an = oracle.analyse(pr)
    An = torch.from_numpy(oracle.a_native(pr.A, np.float64)).cuda()
    Xn = torch.from_numpy(oracle.to_native(X, np.float64)).cuda()
    dS = torch.from_numpy(an["starts"].view(np.int32)).cuda()
    dP = torch.from_numpy(an["pairs"].view(np.int32)).cuda()
    Y = torch.zeros_like(Xn)
    with T.Solver() as s:
        def mult(x, y):
            assert T.lib.tfqmrgpuExt_multiply(s.handle, b"z", 16, 16, pr.nnzbX, dS.data_ptr(), dP.data_ptr(),
                                              An.data_ptr(), x.data_ptr(), y.data_ptr()) == 0
            torch.cuda.synchronize()
        mult(Xn, Y)
        R = oracle.from_native(Y.cpu().numpy())
        R[an["subset"]] -= pr.B
        col = an["colindx"].astype(np.int64)
        res2 = np.zeros((an["nCols"], pr.LN))
        np.add.at(res2, col, (np.abs(R) ** 2).sum(axis=1))
        b2 = np.zeros((an["nCols"], pr.LN))
        np.add.at(b2, col[an["subset"]], (np.abs(pr.B) ** 2).sum(axis=1))
        assert np.sqrt((res2 / b2).max()) <= 1e-9
        # the solver's own figure, against this re-computation with another summation order (rounding of A x at |r| / |b| = 1e-10: ~1e-6)
        assert abs(np.sqrt((res2 / b2).max()) - info["residual"]) <= 1e-4 * info["residual"]
        Z = torch.randn_like(Xn)
        Y2, Y3 = torch.zeros_like(Xn), torch.zeros_like(Xn)
        mult(Z, Y2)
        mult(2.5 * Xn + Z, Y3)
        assert (Y3 - (2.5 * Y + Y2)).abs().max().item() <= 1e-11 * Y3.abs().max().item()


def test_edge_single_block_system(oracle):
    # the Fortran example's first case: one 32x32 block (example/tfqmrgpu_Fortran_example.F90:22-46)
    rng = np.random.default_rng(11)
    A = (rng.standard_normal((1, 32, 32)) + 1j * rng.standard_normal((1, 32, 32))) / 8 + 2 * np.eye(32)
    B = rng.standard_normal((1, 32, 32)) + 1j * rng.standard_normal((1, 32, 32))
    pr = T.Problem([0, 1], [0], A, [0, 1], [0], [0, 1], [0], B, None, 1e-10)
    st, X, info = T.solve_problem(pr, "z", max_iterations=300)
    assert st == 0 and np.abs(X[0] - np.linalg.solve(A[0], B[0])).max() < 1e-8


def test_edge_column_gaps_and_offsets(oracle):
    # block columns 3, 7, 20 (gaps are compressed away, tfqmrgpu.cu:254-315) give the result of columns 0, 1, 2
    pr = PR.stencil_2d(5, 4, 8, 8, 3, seed=13, radius=2.2)
    remap = np.array([3, 7, 20])
    gap = T.Problem(pr.rowPtrA, pr.colIndA, pr.A, pr.rowPtrX, remap[pr.colIndX], pr.rowPtrB, remap[pr.colIndB], pr.B, None, 1e-9)
    st0, X0, i0 = T.solve_problem(pr, "z", shadow_mode=T.SHADOW_GLIBC_RAND)
    st1, X1, i1 = T.solve_problem(gap, "z", shadow_mode=T.SHADOW_GLIBC_RAND)
    assert st0 == st1 == 0 and i0["iterations"] == i1["iterations"] and np.array_equal(X0, X1)
    with T.Solver() as s:
        s.create_plan(gap)
        assert list(s.plan_view()["original_bsrColIndX"]) == [3, 7, 20]


def test_edge_zero_right_hand_side_column(oracle):
    # one RHS of a block column is identically zero: |b| = 0 -> tau = 0 -> that RHS stops with status -3
    # (tfqmrgpu_linalg.hxx:209-212), its bound is NaN and ignored by the max (tfqmrgpu_core.hxx:243); the
    # other right-hand sides converge as usual.  The HIP path must take the same decisions as the oracle.
    pr = PR.stencil_2d(4, 4, 4, 8, 2, seed=17, radius=2.0)
    pr.B[:, :, 5] = 0
    st, X, info = T.solve_problem(pr, "z", threshold=1e-9, max_iterations=100, shadow_mode=T.SHADOW_GLIBC_RAND)
    st0, X0, info0 = oracle.solve(pr, "z", threshold=1e-9, max_iterations=100)
    assert st == st0 and info["iterations"] == info0["iterations"]
    keep = [j for j in range(8) if j != 5]
    assert np.abs(X[:, :, keep] - X0[:, :, keep]).max() <= 1e-7 * np.abs(X0[:, :, keep]).max()
    assert np.array_equal(np.isnan(X[:, :, 5]), np.isnan(X0[:, :, 5]))


def test_edge_rows_without_blocks(torch_cuda, oracle):
    # Y blocks whose pair list is empty must come out as zeros, for every kernel family
    torch = torch_cuda
    for LM, LN in ((16, 16), (8, 8), (4, 5)):
        rng = np.random.default_rng(LM + LN)
        nY = 9
        starts = np.array([0, 0, 2, 2, 2, 5, 5, 5, 5, 6], np.uint32)
        pairs = rng.integers(0, 4, size=12).astype(np.uint32)
        A = rng.uniform(-1, 1, (4, 2, LM, LM)); X = rng.uniform(-1, 1, (nY, 2, LM, LN))
        want = oracle.spmm("z", LM, LN, starts, pairs, A, X)
        dA, dX = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
        dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
        dY = torch.full((nY, 2, LM, LN), 3.0, dtype=torch.float64, device="cuda")
        with T.Solver() as s:
            assert T.lib.tfqmrgpuExt_multiply(s.handle, b"z", LM, LN, nY, dS.data_ptr(), dP.data_ptr(), dA.data_ptr(), dX.data_ptr(), dY.data_ptr()) == 0
            torch.cuda.synchronize()
        got = dY.cpu().numpy()
        assert np.abs(got - want).max() < 1e-12 and np.all(got[[0, 2, 3, 5, 6, 7]] == 0)


def _random_system(rng, LM, LN):
    """random ragged system: A = strongly diagonally dominant random blocks on a random pattern (unsorted rows), X on a
    random set of block columns with gaps in their numbers, ragged per row, B = a random subset of X's blocks that covers
    every column, Fortran or C offsets"""
    mb = int(rng.integers(2, 13))
    ncol = int(rng.integers(1, 5))
    colnames = np.sort(rng.choice(np.arange(0, 12), size=ncol, replace=False))
    rpA, ciA, blocks = [0], [], []
    for r in range(mb):
        others = [c for c in rng.permutation(mb)[: int(rng.integers(0, 4))] if c != r]
        row = list(rng.permutation([r] + others))
        for c in row:
            blk = (rng.uniform(-1, 1, (LM, LM)) + 1j * rng.uniform(-1, 1, (LM, LM))) / (LM * 4)
            if c == r:
                blk = blk + (2.0 + 0.5j) * np.eye(LM)
            blocks.append(blk)
        ciA += row
        rpA.append(len(ciA))
    rpX, ciX = [0], []
    for r in range(mb):
        m = int(rng.integers(0, ncol + 1))
        ciX += list(colnames[rng.permutation(ncol)[:m]])
        rpX.append(len(ciX))
    for c in colnames:                                       # every column at least once
        if c not in ciX:
            r = int(rng.integers(0, mb))
            ciX.insert(rpX[r + 1], int(c))
            for q in range(r + 1, mb + 1):
                rpX[q] += 1
    seen, rpB, ciB = set(), [0], []
    for r in range(mb):
        for c in ciX[rpX[r]:rpX[r + 1]]:
            if c not in seen or rng.random() < 0.4:
                ciB.append(c); seen.add(c)
        rpB.append(len(ciB))
    B = rng.uniform(-1, 1, (len(ciB), LM, LN)) + 1j * rng.uniform(-1, 1, (len(ciB), LM, LN))
    off = int(rng.integers(0, 2))
    return T.Problem(np.array(rpA) + off, np.array(ciA, dtype=np.int64) + off, np.array(blocks), np.array(rpX) + off,
                     np.array(ciX, dtype=np.int64) + off, np.array(rpB) + off, np.array(ciB, dtype=np.int64) + off, B, None, 1e-10, off)


def test_random_ragged_systems(oracle):
    """30 random systems over all block shapes: the GPU solution against the oracle's (same iteration count in double with
    the reference's shadow vector) and against the dense LAPACK solution of the pattern-truncated system"""
    rng = np.random.default_rng(424242)
    for case in range(30):
        LM, LN = SIZES[case % len(SIZES)]
        prec = "z" if case % 3 else "c"
        pr = _random_system(rng, LM, LN)
        tol = 1e-10 if prec == "z" else 1e-4
        st, X, info = T.solve_problem(pr, prec, threshold=tol, max_iterations=300, shadow_mode=T.SHADOW_GLIBC_RAND)
        st0, X0, info0 = oracle.solve(pr, prec, threshold=tol, max_iterations=300)
        assert st == st0 == 0, (case, LM, LN, prec, st, st0)
        scale = np.abs(X0).max()
        assert np.abs(X - X0).max() <= (_tol(prec) if prec == "z" else 1e-3) * scale, (case, LM, LN, prec)
        if prec == "z":
            assert info["iterations"] == info0["iterations"], (case, LM, LN)
            Xd = PR.dense_reference_solution(pr)
            assert np.abs(X - Xd).max() <= 1e-8 * np.abs(Xd).max(), (case, LM, LN)


@pytest.mark.parametrize("prec,size", [("z", (4, 4)), ("z", (4, 8)), ("z", (4, 32)), ("z", (4, 5)), ("c", (4, 4)), ("c", (4, 5)), ("c", (4, 32)), ("c", (8, 9))])
def test_multiply_rows_with_more_products_than_the_index_patch_holds(torch_cuda, oracle, prec, size):
    """k_spmm_m4 and k_spmm_small4 keep the row ranges and index pairs of a chunk in LDS (1024 | 2048 pairs): 70 rows of 38-44 products each are
    more than that, the work groups take their other path (one product at a time, indices from global memory) -- same sums, same order"""
    torch = torch_cuda
    LM, LN = size
    rng = np.random.default_rng(LM * 1000 + LN)
    nY, nA, nX = 70, 31, 45
    starts, pairs = [0], []
    for _ in range(nY):
        for _ in range(int(rng.integers(38, 45))):
            pairs += [int(rng.integers(0, nA)), int(rng.integers(0, nX))]
        starts.append(len(pairs) // 2)
    starts, pairs = np.array(starts, np.uint32), np.array(pairs, np.uint32)
    real = np.float64 if prec == "z" else np.float32
    A = rng.uniform(-1, 1, (nA, 2, LM, LM)).astype(real)
    X = rng.uniform(-1, 1, (nX, 2, LM, LN)).astype(real)
    want = oracle.spmm(prec, LM, LN, starts, pairs, A, np.concatenate([X, np.zeros((nY - nX, 2, LM, LN), real)]))[:nY]
    dA, dX = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
    dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
    dY = torch.full((nY, 2, LM, LN), 7.0, dtype=dA.dtype, device="cuda")
    with T.Solver() as s:
        assert T.lib.tfqmrgpuExt_multiply(s.handle, prec.encode(), LM, LN, nY, dS.data_ptr(), dP.data_ptr(), dA.data_ptr(), dX.data_ptr(), dY.data_ptr()) == 0
        torch.cuda.synchronize()
    eps = 1e-13 if prec == "z" else 2e-5
    assert np.abs(dY.cpu().numpy() - want).max() <= eps * LM * 44, (size, prec)


@pytest.mark.parametrize("prec", ["z", "c"])
@pytest.mark.parametrize("shape", [(16, 16), (8, 8), (4, 5), (4, 4), (4, 8), (4, 32), (8, 9), (32, 32), (16, 64), (32, 64), (64, 64)])
def test_apply_operator_on_plan_data(oracle, prec, shape):
    """X := A*X with the solver's own multiply kernel and block / element order (16 x 16 z: row pairs interleaved)
    against the oracle's product on the caller's order (reference contract: tfqmrgpu_blockmult.hxx:9-93)"""
    LM, LN = shape
    pr = PR.stencil_2d(7, 5, LM, LN, 3, seed=LM + LN + 3, radius=3.1)
    rng = np.random.default_rng(LM * LN)
    X = rng.uniform(-1, 1, (pr.nnzbX, LM, LN)) + 1j * rng.uniform(-1, 1, (pr.nnzbX, LM, LN))
    real = np.float64 if prec == "z" else np.float32
    an = oracle.analyse(pr)
    want = oracle.from_native(oracle.spmm(prec, LM, LN, an["starts"], an["pairs"], oracle.a_native(pr.A, real), oracle.to_native(X, real)))
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(LM, LN, prec))
        s.set_matrix("A", pr.A)
        s.set_matrix("X", X)
        s.apply_operator()
        got = s.get_matrix()
        s.set_matrix("X", X)
        s.apply_operator(3)                       # repetitions start from the same X
        assert np.array_equal(s.get_matrix(), got)
    eps = 1e-13 if prec == "z" else 3e-5
    assert np.abs(got - want).max() <= eps * LM * 6 * max(1.0, np.abs(want).max())


def test_apply_operator_with_column_batches(oracle):
    """8 x 8 complex<double>, three block columns with the same row pattern on a plan of more than 128 chunks: the multiply takes the first two
    columns together and the third alone (k_spmm_ilv8b); X := A*X against the oracle's product, as in the test above"""
    LM = LN = 8
    pr = PR.stencil_2d(48, 30, LM, LN, 3, seed=11)
    rng = np.random.default_rng(5)
    X = rng.uniform(-1, 1, (pr.nnzbX, LM, LN)) + 1j * rng.uniform(-1, 1, (pr.nnzbX, LM, LN))
    an = oracle.analyse(pr)
    want = oracle.from_native(oracle.spmm("z", LM, LN, an["starts"], an["pairs"], oracle.a_native(pr.A, np.float64), oracle.to_native(X, np.float64)))
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(LM, LN, "z"))
        s.set_matrix("A", pr.A)
        s.set_matrix("X", X)
        s.apply_operator()
        got = s.get_matrix()
    assert np.abs(got - want).max() <= 1e-13 * LM * 6 * max(1.0, np.abs(want).max())


def test_three_product_form_is_opt_in(torch_cuda, oracle):
    """32 x 32 complex<double>: the default multiply forms a complex product from four real ones like the reference, so an
    imaginary part 1e-8 times smaller than the real part keeps its digits (up to r02 the three-product form was the default and lost 8
    of them).  With tfqmrgpuExt_setThreeProductMultiply the plan's multiplies use three products: accurate relative to |A||X| (a few
    eps per accumulated term), i.e. the imaginary part may lose the 8 digits -- the documented bound, not more."""
    torch = torch_cuda
    LM = LN = 32
    rng = np.random.default_rng(77)
    nY, nA, nX = 12, 9, 11
    starts, pairs = _rand_pairs(rng, nY, nA, nX, 6)
    A = rng.uniform(-1, 1, (nA, 2, LM, LM)); A[:, 1] *= 1e-8
    X = rng.uniform(-1, 1, (nX, 2, LM, LN)); X[:, 1] *= 1e-8
    Xp = np.concatenate([X, np.zeros((max(0, nY - nX), 2, LM, LN))])
    want = oracle.spmm("z", LM, LN, starts, pairs, A, Xp)[:nY]
    dA, dX = torch.from_numpy(A).cuda(), torch.from_numpy(X).cuda()
    dS, dP = torch.from_numpy(starts.view(np.int32)).cuda(), torch.from_numpy(pairs.view(np.int32)).cuda()
    dY = torch.zeros((nY, 2, LM, LN), dtype=torch.float64, device="cuda")
    with T.Solver() as s:
        assert T.lib.tfqmrgpuExt_multiply(s.handle, b"z", LM, LN, nY, dS.data_ptr(), dP.data_ptr(), dA.data_ptr(), dX.data_ptr(), dY.data_ptr()) == 0
        torch.cuda.synchronize()
    got = dY.cpu().numpy()
    scale = 6 * LM * 1.0                                  # |A||X| summed over at most 6 products of 32 terms, entries <= 1
    eps = np.finfo(np.float64).eps
    assert np.abs(got[:, 0] - want[:, 0]).max() <= 8 * eps * scale
    assert np.abs(got[:, 1] - want[:, 1]).max() <= 16 * eps * scale * 1e-8     # four products: Im accurate relative to ITSELF

    # the option, on a plan: same system solved with and without it
    pr = PR.stencil_2d(8, 8, 32, 32, 2, seed=3, points=13)
    pr.A = pr.A.real + 1e-8j * pr.A.imag                                       # imaginary parts 1e-8 of the real parts
    res = {}
    xin = rng.uniform(-1, 1, (pr.nnzbX, LM, LN)) + 1e-8j * rng.uniform(-1, 1, (pr.nnzbX, LM, LN))
    for on in (False, True):
        with T.Solver() as s:
            s.create_plan(pr)
            s.set_buffer(nbytes=s.buffer_size(LM, LN, "z"))
            s.set_three_product_multiply(on)
            s.set_matrix("A", pr.A)
            s.set_matrix("X", xin)
            s.apply_operator()
            res[on] = s.get_matrix()
    ref = res[False]
    assert np.abs(res[True].real - ref.real).max() <= 64 * eps * 13 * LM * np.abs(ref.real).max()
    d_im = np.abs(res[True].imag - ref.imag).max()
    assert 0 < d_im <= 64 * eps * 13 * LM * np.abs(ref.real).max()             # differs (it IS another formula), within eps |A||X|
    st0, X0, i0 = T.solve_problem(pr, "z", threshold=1e-9, max_iterations=100)
    st1, X1, i1 = T.solve_problem(pr, "z", threshold=1e-9, max_iterations=100, three_products=True)
    assert st0 == st1 == 0 and i0["iterations"] == i1["iterations"]
    assert np.abs(X1 - X0).max() <= 1e-9 * np.abs(X0).max()


def test_profile_keeps_the_first_iteration_apart():
    """tfqmrgpuExt_getProfileFirst: of the launches that did work, those of the first iteration (one per class and solve)"""
    pr = load_problem("fd_16x16_2d")
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(16, 16, "z"))
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        s.set_profiling(1)
        assert s.solve(1e-9, 2000) == 0
        it = s.get_info()["iterations"]
        allp, first = s.profile(), s.profile(first=True)
        for k in ("xpay_v6", "spmm_v4_dot", "x_v6_v7", "spmm_v5_nrm_dot"):
            assert allp[k][0] == it and first[k][0] == 1 and 0 < first[k][1] < allp[k][1], k
        assert s.solve(1e-9, 1) == 9                       # a solve of one iteration: every working launch is a first-iteration launch
        assert s.profile()["spmm_v4_dot"][0] == s.profile(first=True)["spmm_v4_dot"][0] == 1
