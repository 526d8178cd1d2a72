"""User-defined linear operator (include/tfqmrgpu_ext.h section 5), the C-ABI counterpart of the reference's
`action_t` concept (README.md:110-117, tfqmrgpu_blocksparse.hxx:71-199, call site tfqmrgpu_core.hxx:134).
Needs an MI355X: run with `pytest -m gpu`."""
import ctypes as C

import numpy as np
import pytest

import tfqmrgpu_amd as T
from conftest import load_problem
from tfqmrgpu_amd import problems as PR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a GPU; there is no CPU fallback"
    torch.cuda.set_device(0)
    return torch


class _DevArray:
    """wraps a raw device pointer so that torch can see it (no copy)"""
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr, data=(int(ptr), False), version=2)


def _native_A(torch, pr, prec):
    real = torch.float64 if prec == "z" else torch.float32
    At = pr.A.transpose(0, 2, 1)      # the multiply expects A blocks transposed, [k][i]
    return torch.from_numpy(np.ascontiguousarray(np.stack([At.real, At.imag], axis=1))).to(real).cuda()


CASES = [("fd_16x16_small", "z"), ("fd_16x16_small", "c"), ("julia_kat", "z"), ("stencil_8x8", "z"), ("stencil_8x32", "c"),
         ("dense_random_rect", "z"), ("st32x32", "c")]    # st32x32: the quad-interleaved 32 x 32 complex<float> plan (k_spmm_ilvf)


def _problem(name):
    return PR.stencil_2d(6, 6, 32, 32, 2, seed=3) if name == "st32x32" else load_problem(name)


@pytest.mark.parametrize("name,prec", CASES)
def test_operator_that_repeats_the_builtin_multiply(torch_cuda, name, prec):
    # the callback computes the same block-sparse product with tfqmrgpuExt_multiply on the caller's block order:
    # the un-fused schedule must take the same path through the iteration as the fused one
    torch = torch_cuda
    pr = _problem(name)
    tol = pr.tolerance if prec == "z" else 1e-4
    st0, X0, info0 = T.solve_problem(pr, prec, threshold=tol)
    with T.Solver() as s:
        s.create_plan(pr)
        view = s.plan_view()
        s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, prec))
        s.set_matrix("A", 0 * pr.A)            # the values of the built-in operator are not used
        s.set_matrix("B", pr.B)
        An = _native_A(torch, pr, prec)
        dS = torch.from_numpy(view["starts"].view(np.int32)).cuda()
        dP = torch.from_numpy(view["pairs"].view(np.int32)).cuda()
        colindx = view["colindx"].copy()
        seen = []

        def multiply(y, x, cols, nnzbX, nCols, lm, ln, precision, stream):
            if not seen:                       # what the reference hands to action_t::multiply
                got = torch.as_tensor(_DevArray(cols, (nnzbX,), "<u2"), device="cuda").cpu().numpy()
                seen.append((np.array_equal(got, colindx), nnzbX, nCols, lm, ln, precision))
            T._check(T.lib.tfqmrgpuExt_multiply(s.handle, precision.encode(), lm, ln, nnzbX, dS.data_ptr(), dP.data_ptr(),
                                                An.data_ptr(), x, y), "tfqmrgpuExt_multiply")
            return view["nPairs"] * 8.0 * lm * lm * ln

        s.set_operator(multiply)
        st = s.solve(tol, 2000)
        info, X = s.get_info(), s.get_matrix()
        assert seen == [(True, pr.nnzbX, view["nCols"], pr.LM, pr.LN, prec)]
        assert st == st0 == 0
        assert abs(info["iterations"] - info0["iterations"]) <= (1 if prec == "z" else max(3, info0["iterations"] // 3))
        assert info["residual"] <= tol
        assert np.abs(X.astype(np.complex128) - X0).max() <= (1e-7 if prec == "z" else 1e-3) * np.abs(X0).max()
        if info["iterations"] == info0["iterations"]:
            assert info["flops"] == pytest.approx(info0["flops"], rel=1e-12)    # the operator's own count is used
        # back to the built-in operator on the same plan
        s.set_operator(None)
        s.set_matrix("A", pr.A)
        assert s.solve(tol, 2000) == 0
        assert s.get_info()["iterations"] == info0["iterations"] and np.array_equal(s.get_matrix(), X0)


def test_matrix_free_operator(torch_cuda):
    # an operator that is not block-sparse data at all: A = diag(d) with one complex number per (block row, row
    # inside the block), applied with torch on the vectors the solver hands over; the solution is B / d
    torch = torch_cuda
    pr = PR.stencil_2d(6, 5, 8, 32, ncols=3, seed=11)
    mb, LM, LN = pr.mb, pr.LM, pr.LN
    d = (2.0 + PR.hashed_uniform(5, (mb, LM))) + 1j * PR.hashed_uniform(6, (mb, LM))
    rows = np.repeat(np.arange(mb), np.diff(pr.rowPtrX))          # block row of every X block, caller's order
    dX = torch.from_numpy(d[rows]).cuda()                            # [nnzbX][LM] complex128
    eye = T.Problem(np.arange(mb + 1, dtype=np.int32), np.arange(mb, dtype=np.int32), np.zeros((mb, LM, LM), complex),
                    pr.rowPtrX, pr.colIndX, pr.rowPtrB, pr.colIndB, pr.B, tolerance=1e-9)
    calls = []
    with T.Solver() as s:
        s.create_plan(eye)                      # any valid pattern for A: block diagonal
        s.set_buffer(nbytes=s.buffer_size(LM, LN, "z"))
        s.set_matrix("A", eye.A)
        s.set_matrix("B", eye.B)

        def multiply(y, x, cols, nnzbX, nCols, lm, ln, precision, stream):
            X = torch.as_tensor(_DevArray(x, (nnzbX, 2, lm, ln), "<f8"), device="cuda")
            Y = torch.as_tensor(_DevArray(y, (nnzbX, 2, lm, ln), "<f8"), device="cuda")
            dr, di = dX.real[:, :, None], dX.imag[:, :, None]
            Y[:, 0] = dr * X[:, 0] - di * X[:, 1]
            Y[:, 1] = dr * X[:, 1] + di * X[:, 0]
            calls.append(1)
            return 8.0 * nnzbX * lm * ln

        s.set_operator(multiply)
        assert s.solve(1e-9, 100) == 0
        info, X = s.get_info(), s.get_matrix()
    assert info["residual"] <= 1e-9 and info["iterations"] <= 40
    assert len(calls) >= 2 * info["iterations"] + 1                 # two products per iteration + the probes
    want = np.zeros_like(X)
    # B's blocks sit in X at `subset`; every other block of the solution is zero
    pos = {(r, c): k for k, (r, c) in enumerate(zip(rows, pr.colIndX))}
    rowsB = np.repeat(np.arange(mb), np.diff(pr.rowPtrB))
    for b, (r, c) in enumerate(zip(rowsB, pr.colIndB)):
        want[pos[(r, c)]] = pr.B[b] / d[r][:, None]
    assert np.abs(X - want).max() <= 1e-8 * np.abs(want).max()


def test_operator_error_aborts_the_solve(torch_cuda):
    pr = load_problem("julia_kat")
    with T.Solver() as s:
        s.create_plan(pr)
        s.set_buffer(nbytes=s.buffer_size(pr.LM, pr.LN, "z"))
        s.set_matrix("A", pr.A)
        s.set_matrix("B", pr.B)
        cb = T.OPERATOR_CB(lambda *a: 14 + 1000 * 77)
        assert T.lib.tfqmrgpuExt_setOperator(s.plan, cb, None) == 0
        st = T.lib.tfqmrgpu_bsrsv_solve(s.handle, s.plan, 1e-9, 50)
        assert T.decode(st)[:2] == (14, 77)
