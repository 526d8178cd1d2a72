"""the finite-difference example generator and the <LinearProblem> readers.

The generator (tfqmrgpu_amd/fd_generator.py) must write the bytes that the reference's
example/tfqmrgpu_generate_FD_example.cxx:303-883 writes: the four XML files under tests/golden/ ARE outputs of
the reference generator (tests/golden/make_golden.py), two of them pinned by md5 in SURVEY.md Appendix E.
The readers (Python: tfqmrgpu_amd/problems.py, C++: tfqmrgpu_amd/csrc/bench_tfqmrgpu.cpp) follow the schema of
tfQMRgpu/include/tfqmrgpu_example_xml_reader.hxx:125-292 (RowStart or NonzerosPerRow, optional Indirection,
real or complex tensors, scale)."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from tfqmrgpu_amd import problems as PR
from tfqmrgpu_amd.fd_generator import FDExample, main as fd_main

# generate_FD_example arguments: rsb rtb block_edge dimension energy [reference echo] nFD  (tests/golden/make_golden.py)
FIXTURES = {
    "fd_8x8_3d": (1.75, 6.75, 2, 3, 0.0, 4),
    "fd_16x16_2d": (6, 24, 4, 2, -0.25, 4),
    "fd_16x16_small": (4, 12, 4, 2, -0.25, 4),
    "fd_4x4_2d": (3, 9, 2, 2, -0.1, 4),
}
MD5 = {
    "fd_8x8_3d": ("8bbd6b1fda2c267fa7e5d49aaf5c67a6", 13945),
    "fd_16x16_2d": ("fa4f8c448cb95900df3870cfe5eb150b", 12643),
}
EXE = os.path.join(ROOT, "tfqmrgpu_amd", "lib", "bench_tfqmrgpu")


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_generator_writes_the_reference_bytes(name):
    want = open(os.path.join(GOLDEN, name + ".xml"), "rb").read()
    got = FDExample(*FIXTURES[name]).to_xml().encode()
    assert got == want
    if name in MD5:
        assert (hashlib.md5(got).hexdigest(), len(got)) == MD5[name]


def test_generator_cli_matches_the_reference_cli(tmp_path):
    # positional arguments of the reference's main (tfqmrgpu_generate_FD_example.cxx:915-923), output name FD_problem.xml (:312)
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        fd_main(["6", "24", "4", "2", "-0.25", "n", "3", "4"])
    finally:
        os.chdir(cwd)
    assert open(tmp_path / "FD_problem.xml", "rb").read() == open(os.path.join(GOLDEN, "fd_16x16_2d.xml"), "rb").read()


def test_generator_problem_equals_what_the_reader_returns():
    for name, args in FIXTURES.items():
        a, b = FDExample(*args).problem(), PR.read_xml(os.path.join(GOLDEN, name + ".xml"))
        for k in ("rowPtrA", "colIndA", "rowPtrX", "colIndX", "rowPtrB", "colIndB"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), (name, k)
        assert np.array_equal(a.A, b.A) and np.array_equal(a.B, b.B) and a.tolerance == b.tolerance


def test_bench_instance_sizes():
    # P2 = generate_FD_example 16 120 4 2 -0.25: the instance bench.py times
    ex = FDExample(16, 120, 4, 2, -0.25, 4)
    assert (ex.nrows, len(ex.colIndA), len(ex.colIndX), len(ex.colIndB), ex.n_sources) == (3573, 17589, 138229, 49, 49)
    import tfqmrgpu_amd as T
    with T.Solver() as s:                       # createPlan is host-only: the pair count of the analysis
        s.create_plan(ex.problem())
        v = s.plan_view()
    assert (v["nPairs"], v["nCols"]) == (679189, 49)


# ---- readers ---------------------------------------------------------------------------------------
def _xml(rowtag, indirection, complex_tensor, scale):
    """2 block rows, 2x3 blocks; A = 3 blocks drawn from 2 stored ones through an indirection list (or 3 stored ones)"""
    rng = np.random.default_rng(5)
    nstored = 2 if indirection else 3
    ind = [1, 0, 1] if indirection else [0, 1, 2]
    data = rng.integers(-9, 9, size=(nstored, 2, 2, 2 if complex_tensor else 1)).astype(float)
    rows = ('<RowStart rows="2">0 2 3</RowStart>' if rowtag == "RowStart" else '<NonzerosPerRow rows="2">2 1</NonzerosPerRow>')
    a = ['<?xml version="1.0"?>\n<LinearProblem problem_kind="A*X==B" tolerance="2.500e-07">\n <!-- a comment -->\n',
         '<BlockSparseMatrix id="A"><SparseMatrix type="CSR"><CompressedSparseRow>%s' % rows,
         '<ColumnIndex nonzeros="3">0 1\n 1</ColumnIndex></CompressedSparseRow>']
    if indirection:
        a.append('<Indirection nonzeros="3">%s</Indirection>' % " ".join(map(str, ind)))
    a.append('</SparseMatrix><DataTensor type="%s" rank="3" dimensions="%d 2 2"%s>%s</DataTensor></BlockSparseMatrix>\n'
             % ("complex" if complex_tensor else "real", nstored, (' scale="%.16e"' % scale) if scale != 1 else "",
                " ".join("%.15g" % v for v in data.reshape(-1))))
    bx = rng.integers(-9, 9, size=(2, 2, 3)).astype(float)
    for name in "BX":
        a.append('<BlockSparseMatrix id="%s"><SparseMatrix type="CSR"><CompressedSparseRow><NonzerosPerRow rows="2">1 1</NonzerosPerRow>'
                 '<ColumnIndex nonzeros="2">0 0</ColumnIndex></CompressedSparseRow></SparseMatrix>' % name)
        if name == "B":
            a.append('<DataTensor type="real" rank="3" dimensions="2 2 3">%s</DataTensor>' % " ".join("%.15g" % v for v in bx.reshape(-1)))
        else:
            a.append('<DataTensor type="real" rank="3" dimensions="0 2 3"></DataTensor>')
        a.append("</BlockSparseMatrix>\n")
    a.append("</LinearProblem>\n")
    blocks = (data[..., 0] + 1j * data[..., 1]) if complex_tensor else data[..., 0].astype(complex)
    return "".join(a), blocks[ind] * scale, bx.astype(complex)


READER_CASES = [("RowStart", False, False, 1.0), ("NonzerosPerRow", True, False, 0.125), ("RowStart", True, True, 3.0),
                ("NonzerosPerRow", False, True, 1.0)]


@pytest.mark.parametrize("rowtag,indirection,cplx,scale", READER_CASES)
def test_python_reader_schema(tmp_path, rowtag, indirection, cplx, scale):
    text, A, B = _xml(rowtag, indirection, cplx, scale)
    path = tmp_path / "p.xml"
    path.write_text(text)
    pr = PR.read_xml(str(path))
    assert pr.tolerance == 2.5e-7 and (pr.mb, pr.LM, pr.LN) == (2, 2, 3)
    assert list(pr.rowPtrA) == [0, 2, 3] and list(pr.colIndA) == [0, 1, 1]
    assert list(pr.rowPtrX) == [0, 1, 2] and list(pr.colIndB) == [0, 0]
    assert np.array_equal(pr.A, A) and np.array_equal(pr.B, B)
    assert pr.X.shape == (2, 2, 3) and not pr.X.any()


@pytest.mark.parametrize("rowtag,indirection,cplx,scale", READER_CASES)
def test_cpp_reader_schema(tmp_path, rowtag, indirection, cplx, scale):
    # `bench_tfqmrgpu read <file>` parses with the C++ reader and prints what it found (no GPU involved)
    text, A, B = _xml(rowtag, indirection, cplx, scale)
    path = tmp_path / "p.xml"
    path.write_text(text)
    r = subprocess.run([EXE, "read", str(path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout
    assert re.search(r"# found tolerance= 2.5e-07", out)
    got = {}
    for m in re.finditer(r"^# operator (\w) rows (\d+) nnzb (\d+) block (\d+) x (\d+) rowPtr \[([^\]]*)\] colInd \[([^\]]*)\] checksum (\S+) (\S+)$", out, re.M):
        got[m.group(1)] = m.groups()[1:]
    assert set(got) == {"A", "B", "X"}
    assert got["A"][:4] == ("2", "3", "2", "2") and got["A"][4].split() == ["0", "2", "3"] and got["A"][5].split() == ["0", "1", "1"]
    w = np.arange(1, A.size + 1).reshape(A.shape)      # position-weighted checksum: catches a transposed or mis-indirected block
    assert complex(float(got["A"][6]), float(got["A"][7])) == pytest.approx((A * w).sum(), rel=1e-13, abs=1e-12)
    wb = np.arange(1, B.size + 1).reshape(B.shape)
    assert complex(float(got["B"][6]), float(got["B"][7])) == pytest.approx((B * wb).sum(), rel=1e-13, abs=1e-12)
    assert got["X"][1] == "2" and float(got["X"][6]) == 0.0


def test_cpp_reader_on_the_golden_files():
    for name in FIXTURES:
        pr = PR.read_xml(os.path.join(GOLDEN, name + ".xml"))
        r = subprocess.run([EXE, "read", os.path.join(GOLDEN, name + ".xml")], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"^# operator A rows (\d+) nnzb (\d+) block (\d+) x (\d+) rowPtr \[([^\]]*)\] colInd \[([^\]]*)\] checksum (\S+) (\S+)$", r.stdout, re.M)
        assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (pr.mb, pr.nnzbA, pr.LM)
        w = np.arange(1, pr.A.size + 1).reshape(pr.A.shape)
        assert float(m.group(7)) == pytest.approx(float((pr.A.real * w).sum()), rel=1e-12)
