"""The CPU oracle (oracle/tfqmr_oracle.c) against everything that pins it:
golden vectors recorded from the compiled reference, the md5 pins of SURVEY.md Appendix E, the
reference's own known-answer tests, and (in the build container) the live reference library."""
import hashlib

import numpy as np
import pytest

from conftest import ALL_NAMES, golden_solves, load_golden, load_problem, offset1
from tfqmrgpu_amd import problems as PR

PLAN_KEYS = ("pairs", "starts", "subset", "colindx")


@pytest.mark.parametrize("name", ALL_NAMES)
def test_analysis_bit_exact(oracle, name):
    pr, g = load_problem(name), load_golden(name)
    an = oracle.analyse(pr)
    assert an["status"] == 0
    assert an["nCols"] == int(g["plan_nCols"])
    for k in PLAN_KEYS + ("original_bsrColIndX",):
        assert np.array_equal(an[k], g["plan_" + k]), k
    an1 = oracle.analyse(offset1(pr))  # Fortran indices: same lists, raw column numbers kept
    for k in PLAN_KEYS:
        assert np.array_equal(an1[k], g["plan_" + k]), k
    assert np.array_equal(an1["original_bsrColIndX"], g["plan_original_bsrColIndX_off1"])


def _md5(a):
    return hashlib.md5(("".join("%d\n" % v for v in a)).encode()).hexdigest()


def test_analysis_md5_pins_of_survey(oracle):
    # md5 over a text dump (one decimal integer per line) of the reference's lists
    pins = {
        "fd_8x8_3d": dict(pairs="edeea92e968b4b33dab73f663638f63d", starts="e74b690274d0c17f0fb7146402fff739",
                          subset="897316929176464ebc9ad085f31e7284", colindx="0ed700da0ff3acb5336070d9b187be09"),
        "fd_16x16_2d": dict(pairs="bada96c6bed0b37f873dec1c2642e483", starts="3c8afde5b7d80af04e4f557b68000ab3",
                            subset="834da8738803cef84a3ec705130e4994", colindx="d83347d6fe29031116cf546cd949634b"),
    }
    for name, pin in pins.items():
        an = oracle.analyse(load_problem(name))
        for k, h in pin.items():
            assert _md5(an[k]) == h, (name, k)


@pytest.mark.parametrize("name", ALL_NAMES)
def test_solve_matches_reference_golden(oracle, name):
    pr, g = load_problem(name), load_golden(name)
    for prec, tol, maxit in golden_solves(g):
        st, X, info = oracle.solve(pr, prec, threshold=tol, max_iterations=maxit)
        tag = "solve_%s_" % prec
        assert st == int(g[tag + "status"])
        assert info["iterations"] == int(g[tag + "iterations"])
        assert info["flops"] == float(g[tag + "flops"])
        assert abs(info["residual"] - float(g[tag + "residual"])) <= 1e-9 * float(g[tag + "residual"])   # (not pytest.approx: abs 1e-12)
        scale = float(g[tag + "maxabsX"])
        eps = 1e-12 if prec == "z" else 1e-5
        if tag + "X" in g:
            assert np.abs(X - g[tag + "X"]).max() <= eps * scale
        else:
            assert np.abs(X.reshape(-1)[::97] - g[tag + "X_sample"]).max() <= eps * scale
            assert abs(X.sum() - complex(g[tag + "sumX"])) <= (1e-9 if prec == "z" else 1e-5) * abs(complex(g[tag + "sumX"])) + 1e-12


def test_julia_known_answer(oracle):
    # example/tfqmrgpu_Julia_example.jl:117-120: the solution is the straight line k/8, per-RHS phase i^p
    pr = PR.julia_kat()
    st, X, info = oracle.solve(pr, "z", threshold=1.2e-8, max_iterations=210)
    assert st == 0 and info["iterations"] == 7
    line = np.arange(1, 8) / 8.0
    for j in range(5):
        assert np.abs(X[:, j % 4, j] - line * 1j ** (j // 4)).max() < 1e-12
    assert np.abs(np.round(X * 8) - X * 8).max() / 8 < 1e-12


def test_true_system_is_solved(oracle):
    # the Fortran example's own check (example/tfqmrgpu_Fortran_example.F90:108-126): A*X == B for a
    # non-symmetric dense system -- fails if the blocks of A are taken transposed
    pr = PR.dense_random()
    st, X, _ = oracle.solve(pr, "z", threshold=1e-10, max_iterations=500)
    assert st == 0
    Xd = PR.dense_reference_solution(pr)
    assert np.abs(X - Xd).max() < 1e-8 * np.abs(Xd).max()


def test_glibc_rand_restatement(oracle):
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    want = np.array([libc.rand() for _ in range(2000)], dtype=np.int64)
    got = oracle.shadow_glibc(2000)
    denom = np.float32(1.0 / 2147483647)
    assert np.array_equal(got, (want.astype(np.float32) * denom).astype(np.float32))
    assert abs(float(got[0]) - 0.840188) < 1e-6  # first value of the never-seeded sequence


def test_against_live_reference(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not present (it can only be built where /root/reference exists; the built files travel with gpurun snapshots)")
    ref = oracle.Reference()
    pr = PR.stencil_2d(5, 5, 4, 8, 3, seed=77, radius=2.0)
    a, r = oracle.analyse(pr), ref.analyse(pr)
    for k in PLAN_KEYS + ("original_bsrColIndX",):
        assert np.array_equal(a[k], r[k])
    for prec, tol in (("z", 1e-9), ("c", 1e-4)):
        st, X, info = oracle.solve(pr, prec, threshold=tol, max_iterations=200)
        st2, X2, info2 = ref.solve_staged(pr, prec, threshold=tol, max_iterations=200)
        assert (st, info["iterations"]) == (st2, info2["iterations"])
        assert info["flops"] == info2["flops"]
        assert np.abs(X - X2).max() <= (1e-12 if prec == "z" else 1e-5) * np.abs(X2).max()


def test_state_dump_hook(oracle):
    # the test hook behind tests/test_gpu_hash_mode.py::test_work_vectors_after_k_iterations...: the dumped vectors are the
    # driver's own (x of the dump = the X handed back; v9 = A v6 - (v6 changed since) is not checkable, v8 = A v6 is)
    pr = load_problem("fd_16x16_2d")
    st, X, info = oracle.solve(pr, "z", threshold=1e-30, max_iterations=3, dump_iteration=3)
    st2, X2, info2 = oracle.solve(pr, "z", threshold=1e-30, max_iterations=3)
    assert st == st2 == 9 and np.array_equal(X, X2) and "vectors" not in info2
    v = info["vectors"]
    assert sorted(v) == [1, 4, 5, 6, 7, 8, 9] and np.array_equal(v[1], X)
    an = oracle.analyse(pr)
    real = np.float64
    Y = oracle.spmm("z", pr.LM, pr.LN, an["starts"], an["pairs"], oracle.a_native(pr.A, real), oracle.to_native(v[6], real))
    assert np.array_equal(oracle.from_native(Y), v[8])        # v8 = A v6 (tfqmrgpu_core.hxx:224), same routine, same bits
