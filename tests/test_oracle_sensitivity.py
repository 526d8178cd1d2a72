"""How much of the GPU-against-oracle deviation is the FIXTURE's own sensitivity to rounding?

The tfQMR recurrences amplify rounding differences; by how much depends on the system (iteration count, conditioning).  This test
measures that on the CPU alone: the oracle against the same oracle with two other, equally valid roundings of the block multiply
  * `fma`     : compiled with -mfma -ffp-contract=fast (products fused into the sums, as the GPU kernels do explicitly),
  * `running` : ONE running sum over all block products of a Y element (the order VERDICT r02 asked about for the 4-row kernel;
                reference and oracle add a per-product sum to the block, tfqmrgpu_blocksparse.hxx:160-177),
and holds the per-fixture tolerances of tests/test_gpu_parity.py (2 x what was observed on MI355X) against it.  Result (r03,
profiles/r03_sensitivity.txt): on fd_4x4_2d (32 iterations) the oracle differs from ITSELF by 5.8e-6 in the final residual and
1.5e-5 in the bound history when only the contraction of the products changes -- the same size as the GPU kernel's deviation
(3.8e-5 / 9.2e-6).  The deviation of the 4-row shapes is the fixture's conditioning, not the summation order of k_spmm_small4
(which is the oracle's: a per-product sum added to the block); X itself agrees to 2e-10 in every variant."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_problem
from tolerances import Z_TOL

FIXTURES = ["fd_4x4_2d", "fd_16x16_2d", "fd_16x16_small", "dense_random", "stencil_8x8", "stencil_8x32", "fd_8x8_3d", "julia_kat"]


def _have_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read().replace("\n", " ")
    except OSError:
        return False


@pytest.fixture(scope="module")
def variants(tmp_path_factory, oracle):
    if not _have_fma():
        pytest.skip("host CPU without FMA instructions")
    out = tmp_path_factory.mktemp("oracle_variants")
    src = os.path.join(ROOT, "oracle", "tfqmr_oracle.c")
    libs = {}
    for name, flags in (("fma", ["-mfma", "-ffp-contract=fast"]), ("running", ["-DTFQO_RUNNING_SUM"])):
        so = str(out / ("liboracle_%s.so" % name))
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fopenmp", "-fPIC", "-shared"] + flags + ["-o", so, src, "-lm"])
        libs[name] = so
    return libs


def _solve_with(oracle, so, pr):
    keep = (oracle._lib, oracle.ORACLE_SO)
    try:
        oracle._lib, oracle.ORACLE_SO = None, so
        oracle.lib()
        return oracle.solve(pr, "z", threshold=pr.tolerance, max_iterations=500)
    finally:
        oracle._lib, oracle.ORACLE_SO = keep


def deviations(oracle, libs, name):
    pr = load_problem(name)
    st0, X0, i0 = oracle.solve(pr, "z", threshold=pr.tolerance, max_iterations=500)
    h0 = np.array(i0["bound_history"])
    worst = dict(hist=0.0, half=0.0, res=0.0, x=0.0)
    for so in libs.values():
        st, X, i = _solve_with(oracle, so, pr)
        assert st == st0 == 0 and i["iterations"] == i0["iterations"]
        h = np.array(i["bound_history"])
        half = (len(h) + 1) // 2
        worst["hist"] = max(worst["hist"], float(np.abs(h / h0 - 1).max()))
        worst["half"] = max(worst["half"], float(np.abs(h[:half] / h0[:half] - 1).max()))
        worst["res"] = max(worst["res"], abs(i["residual"] / i0["residual"] - 1))
        worst["x"] = max(worst["x"], float(np.abs(X - X0).max() / np.abs(X0).max()))
    return worst, i0["iterations"]


@pytest.mark.parametrize("name", FIXTURES)
def test_gpu_tolerances_are_the_fixtures_own_sensitivity(oracle, variants, name):
    w, it = deviations(oracle, variants, name)
    t = Z_TOL[name]
    # a tolerance of the GPU comparison (2 x observed there) may exceed what two CPU roundings of the same algorithm differ by
    # by one order of magnitude at most: anything looser would hide a real defect, anything the CPU variants already exceed is noise
    assert t["res"] <= 30 * max(w["res"], 1e-7), (name, t["res"], w)
    assert t["hist"] <= 30 * max(w["hist"], 1e-11), (name, t["hist"], w)
    assert w["x"] <= 1e-9, (name, w)          # the SOLUTION is insensitive: every variant agrees far inside the 1e-7 of the parity tests


def test_four_row_fixture_is_the_sensitive_one(oracle, variants):
    w4, it4 = deviations(oracle, variants, "fd_4x4_2d")
    w16, it16 = deviations(oracle, variants, "fd_16x16_2d")
    assert it4 == 32 and it16 == 13
    # 32 iterations amplify a changed rounding of the multiply to ~1e-5 in the history and ~5e-6 in the residual on the CPU alone,
    # 13 iterations to ~1e-10 / ~1e-7: the GPU deviations (9.2e-6 / 3.8e-5 and 1.3e-10 / 1.1e-7) are of exactly that size
    assert 1e-6 < w4["hist"] < 1e-3 and 5e-7 < w4["res"] < 1e-3
    assert w16["hist"] < 1e-9 and w16["res"] < 3e-6


if __name__ == "__main__":      # python tests/test_oracle_sensitivity.py > profiles/r03_sensitivity.txt
    import sys
    import tempfile
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as O
    O.lib()
    d = tempfile.mkdtemp()
    libs = {}
    for nm, flags in (("fma", ["-mfma", "-ffp-contract=fast"]), ("running", ["-DTFQO_RUNNING_SUM"])):
        libs[nm] = os.path.join(d, "liboracle_%s.so" % nm)
        subprocess.check_call(["gcc", "-O2", "-std=c99", "-fopenmp", "-fPIC", "-shared"] + flags + ["-o", libs[nm], os.path.join(ROOT, "oracle", "tfqmr_oracle.c"), "-lm"])
    print("# oracle against itself with another rounding of the block multiply (fma-contracted | one running sum): worst relative deviation")
    print("# %-18s %4s %10s %10s %10s %10s | GPU tolerance (2 x observed on MI355X): hist half res" % ("fixture", "it", "hist", "first half", "residual", "X"))
    for nm in FIXTURES:
        w, it = deviations(O, libs, nm)
        t = Z_TOL[nm]
        print("  %-18s %4d %10.2e %10.2e %10.2e %10.2e | %8.1e %8.1e %8.1e" % (nm, it, w["hist"], w["half"], w["res"], w["x"], t["hist"], t["half"], t["res"]))
