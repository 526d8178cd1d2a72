"""CPU-side checks of the C-ABI library: it loads, exports every symbol of include/*.h, and its
host logic (createPlan index analysis, status codes, block-size table, column sharding) is
bit-identical with the reference.  No device kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import tfqmrgpu_amd as T
from conftest import ALL_NAMES, ROOT, load_golden, load_problem, offset1
from tfqmrgpu_amd import problems as PR

PLAN_KEYS = ("pairs", "starts", "subset", "colindx")


def test_every_declared_symbol_is_exported():
    declared = set()
    for h in ("tfqmrgpu.h", "tfqmrgpu_ext.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(tfqmrgpu\w*)\s*\((?!\s*\*)", text))   # not the function-pointer typedefs
    assert set(T.EXPORTED_SYMBOLS) <= declared and set(T.EXT_SYMBOLS) <= declared
    for name in sorted(declared) + T.FORTRAN_SYMBOLS:
        assert hasattr(T.lib, name), name
    assert len(T.EXPORTED_SYMBOLS) == 21 and len(T.FORTRAN_SYMBOLS) == 18


@pytest.mark.parametrize("name", ALL_NAMES)
def test_create_plan_bit_exact(name):
    pr, g = load_problem(name), load_golden(name)
    for p, origkey in ((pr, "plan_original_bsrColIndX"), (offset1(pr), "plan_original_bsrColIndX_off1")):
        with T.Solver() as s:
            assert s.create_plan(p) == 0
            v = s.plan_view()
        assert v["nCols"] == int(g["plan_nCols"]) and v["nPairs"] == len(g["plan_pairs"]) // 2
        for k in PLAN_KEYS:
            assert v[k].dtype == g["plan_" + k].dtype and np.array_equal(v[k], g["plan_" + k]), k
        assert np.array_equal(v["original_bsrColIndX"], g[origkey])


def test_unsorted_and_duplicate_columns_first_match_wins(oracle):
    # the reference searches linearly and takes the first hit (bsr.hxx:27-39)
    rng = np.random.default_rng(5)
    mb = 9
    rpA, ciA, rpX, ciX = [0], [], [0], []
    for r in range(mb):
        ciA += list(rng.permutation(mb)[: rng.integers(1, 5)])
        rpA.append(len(ciA))
        cols = list(rng.permutation(7)[: rng.integers(1, 5)] * 3 + 2)
        if r % 4 == 0:
            cols.append(cols[0])  # a duplicate entry in the row
        ciX += cols
        rpX.append(len(ciX))
    rpB, ciB = [0], []
    for r in range(mb):  # B: every (row, column) of X once
        ciB += list(dict.fromkeys(ciX[rpX[r]:rpX[r + 1]]))
        rpB.append(len(ciB))
    A = np.zeros((len(ciA), 4, 4), complex)
    B = np.zeros((len(ciB), 4, 4), complex)
    pr = T.Problem(rpA, ciA, A, rpX, ciX, rpB, ciB, B)
    an = oracle.analyse(pr)
    with T.Solver() as s:
        st = T.lib.tfqmrgpu_bsrsv_createPlan(s.handle, C.byref(s.plan), pr.mb, T._ptr(pr.rowPtrA), pr.nnzbA, T._ptr(pr.colIndA),
                                             T._ptr(pr.rowPtrX), pr.nnzbX, T._ptr(pr.colIndX), T._ptr(pr.rowPtrB), pr.nnzbB,
                                             T._ptr(pr.colIndB), 0, 0)
        assert st == 0 and an["status"] == 0
        v = s.plan_view()
        for k in PLAN_KEYS + ("original_bsrColIndX",):
            assert np.array_equal(v[k], an[k]), k
    # two blocks of B on one block of X: the reference accepts it, the GPU build refuses (code 19)
    dup = T.Problem(rpA, ciA, A, rpX, ciX, rpX, ciX, np.zeros((len(ciX), 4, 4), complex))
    assert oracle.analyse(dup)["status"] == 0
    with T.Solver() as s:
        assert T.decode(_create(dup, s.handle, s.plan))[0] == 19


def _create(pr, handle, plan):
    return T.lib.tfqmrgpu_bsrsv_createPlan(handle, C.byref(plan), pr.mb, T._ptr(pr.rowPtrA), pr.nnzbA, T._ptr(pr.colIndA),
                                           T._ptr(pr.rowPtrX), pr.nnzbX, T._ptr(pr.colIndX), T._ptr(pr.rowPtrB), pr.nnzbB,
                                           T._ptr(pr.colIndB), pr.index_offset, 0)


def test_create_plan_status_codes(oracle):
    pr = PR.stencil_2d(4, 3, 4, 4, 3, seed=9)
    s = T.Solver()
    # B block outside the pattern of X: code 13, the block row travels in the line field (tfqmrgpu.cu:245)
    bad = T.Problem(pr.rowPtrA, pr.colIndA, pr.A, pr.rowPtrX, pr.colIndX, pr.rowPtrB, pr.colIndB + 5, pr.B)
    st = _create(bad, s.handle, s.plan)
    row = int(np.nonzero(np.diff(pr.rowPtrB))[0][0])
    assert st == 13 + 1000 * row and not s.plan
    assert oracle.analyse(bad)["status"] == st
    assert "row %d" % row in T.error_string(st)
    # a column of X without any block of B: code 11, the count travels in the line field (tfqmrgpu.cu:335)
    keep = pr.colIndB != 1
    rpB = np.concatenate([[0], np.cumsum([np.sum(keep[pr.rowPtrB[r]:pr.rowPtrB[r + 1]]) for r in range(pr.mb)])])
    nob = T.Problem(pr.rowPtrA, pr.colIndA, pr.A, pr.rowPtrX, pr.colIndX, rpB, pr.colIndB[keep], pr.B[keep])
    st = _create(nob, s.handle, s.plan)
    assert st == 11 + 1000 * 1 and oracle.analyse(nob)["status"] == st
    # inconsistent nnzb: code 14
    st = T.lib.tfqmrgpu_bsrsv_createPlan(s.handle, C.byref(s.plan), pr.mb, T._ptr(pr.rowPtrA), pr.nnzbA + 1, T._ptr(pr.colIndA),
                                         T._ptr(pr.rowPtrX), pr.nnzbX, T._ptr(pr.colIndX), T._ptr(pr.rowPtrB), pr.nnzbB,
                                         T._ptr(pr.colIndB), 0, 0)
    assert T.decode(st)[0] == 14
    # *plan must be NULL on entry: code 7 (tfqmrgpu.cu:161)
    assert s.create_plan(pr) == 0
    assert T.decode(_create(pr, s.handle, s.plan))[0] == 7
    # getInfo with nothing to fill: 3 (tfqmrgpu.cu:678); bufferSize argument checks (tfqmrgpu.cu:377-379)
    assert T.lib.tfqmrgpu_bsrsv_getInfo(s.handle, s.plan, None, None, None, None) == 3
    n = C.c_size_t(0)
    assert T.decode(T.lib.tfqmrgpu_bsrsv_bufferSize(s.handle, s.plan, 4, 8, 4, 4, b"z", C.byref(n)))[0] == 14
    assert T.decode(T.lib.tfqmrgpu_bsrsv_bufferSize(s.handle, s.plan, 8, 8, 4, 4, b"z", C.byref(n)))[0] == 14
    st = T.lib.tfqmrgpu_bsrsv_bufferSize(s.handle, s.plan, 6, 6, 6, 6, b"z", C.byref(n))
    assert T.decode(st) == (12, 6, 6)  # missing block size: char = LM, line = LN (tfqmrgpu.cu:70)
    assert T.lib.tfqmrgpu_bsrsv_bufferSize(s.handle, s.plan, 4, 4, 4, 4, b"z", C.byref(n)) == 0 and n.value > 0
    nz = n.value
    assert T.lib.tfqmrgpu_bsrsv_bufferSize(s.handle, s.plan, 4, 4, 4, 4, b"f", C.byref(n)) == 0 and n.value < nz
    # setBuffer(NULL): 7; getBuffer before setBuffer: 7
    assert T.decode(T.lib.tfqmrgpu_bsrsv_setBuffer(s.handle, s.plan, None))[0] == 7
    p = C.c_void_p(None)
    assert T.decode(T.lib.tfqmrgpu_bsrsv_getBuffer(s.handle, s.plan, C.byref(p)))[0] == 7
    # argument decoding of setMatrix happens before any device work (tfqmrgpu.cu:481-533)
    a = np.zeros(8)
    assert T.decode(T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"A", T._ptr(a), b"z", 4, 4, b"n", 0x77)) == (15, 0x77, 0)
    assert T.decode(T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"A", T._ptr(a), b"z", 4, 4, b"q", 0x55))[::2] == (17, ord("q"))
    assert T.decode(T.lib.tfqmrgpu_bsrsv_setMatrix(s.handle, s.plan, b"Q", T._ptr(a), b"z", 4, 4, b"n", 0x55))[::2] == (18, ord("Q"))
    assert T.decode(T.lib.tfqmrgpu_bsrsv_getMatrix(s.handle, s.plan, b"A", T._ptr(a), b"z", 4, 4, b"n", 0x55))[::2] == (14, ord("A"))
    s.close()
    h = C.c_void_p(1)
    assert T.decode(T.lib.tfqmrgpuCreateHandle(C.byref(h)))[0] == 14  # handle must be NULL on entry


def test_allowed_block_sizes_like_the_reference():
    want = [(4, 4), (4, 5), (4, 8), (4, 32), (8, 8), (8, 9), (8, 10), (8, 32), (8, 64),
            (16, 16), (16, 32), (16, 64), (32, 32), (32, 64), (64, 64)]  # allowed_block_sizes.h:4-18
    n = C.c_int32(0)
    arr = (C.c_int32 * 200)(*([7] * 200))
    assert T.lib.tfqmrgpu_bsrsv_allowedBlockSizes(C.byref(n), arr, 200) == 0
    assert n.value == 15 and [(arr[2 * i], arr[2 * i + 1]) for i in range(15)] == want
    assert arr[30] == 7  # the array is only cleared when *number != 0 on entry (tfqmrgpu.cu:83)
    n = C.c_int32(1)
    assert T.lib.tfqmrgpu_bsrsv_allowedBlockSizes(C.byref(n), arr, 200) == 0 and arr[30] == 0
    small = (C.c_int32 * 8)()
    n = C.c_int32(0)
    st = T.lib.tfqmrgpu_bsrsv_allowedBlockSizes(C.byref(n), small, 8)  # pairs stored while 2*n < arrayLength
    assert n.value == 15 and T.decode(st)[0] == 14 and list(small)[:6] == [4, 4, 4, 5, 4, 8]
    for lm, ln in want:
        assert T.lib.tfqmrgpu_bsrsv_blockSizeMissing(lm, ln) == 0
    assert T.decode(T.lib.tfqmrgpu_bsrsv_blockSizeMissing(3, 5)) == (12, 5, 3)


def test_error_strings():
    assert T.error_string(0) == ""
    assert T.error_string(9) == "tfQMRgpu: Max number of iterations exceeded!"
    assert T.error_string(6) == "tfQMRgpu: All components have broken down!"
    assert T.error_string(12 + 1000 * 64 + 10000000 * 16) == "tfQMRgpu: Missing blocksize 16 x 64!"
    assert T.error_string(17 + 1000 * 498 + 10000000 * ord("q")) == "tfQMRgpu: Unknown transposition 'q' at line 498!"
    assert T.error_string(15 + 1000 * 0x77) == "tfQMRgpu: Unknown data layout '0x77'!"
    assert T.error_string(11 + 1000 * 3) == "tfQMRgpu: B has 3 zero columns, will break!"


def test_shard_columns_partition():
    pr = PR.stencil_2d(6, 6, 4, 4, 7, seed=2, radius=2.2)
    cols = np.unique(pr.colIndX)
    for nranks in (1, 2, 3, 7):
        seen_x, seen_b, first = [], [], 0
        for rank in range(nranks):
            sub, xb, bb = T.shard_columns(pr, nranks, rank)
            assert sub.first_col == first and sub.n_cols >= 1
            first += sub.n_cols
            mine = cols[sub.first_col: sub.first_col + sub.n_cols]
            assert np.array_equal(np.unique(sub.colIndX), mine)
            assert np.array_equal(pr.colIndX[xb], sub.colIndX) and np.array_equal(pr.colIndB[bb], sub.colIndB)
            rows = np.repeat(np.arange(pr.mb), np.diff(pr.rowPtrX))
            assert np.array_equal(rows[xb], np.repeat(np.arange(pr.mb), np.diff(sub.rowPtrX)))
            seen_x += list(xb)
            seen_b += list(bb)
        assert first == len(cols)
        assert sorted(seen_x) == list(range(pr.nnzbX)) and sorted(seen_b) == list(range(pr.nnzbB))


def _random_pattern_problem(rng):
    """random ragged patterns: unsorted rows, duplicate entries, gaps in the column numbers of X, rows without
    blocks, B on a random subset of X's blocks with every column covered, C or Fortran index offset"""
    mb = int(rng.integers(1, 24))
    ncol = int(rng.integers(1, 9))
    colnames = np.sort(rng.choice(np.arange(0, 40), size=ncol, replace=False))   # gaps: empty columns in between
    rpA, ciA, rpX, ciX = [0], [], [0], []
    for r in range(mb):
        n = int(rng.integers(0, min(mb, 6) + 1))
        row = list(rng.choice(mb, size=n, replace=False))
        if n and rng.random() < 0.2:
            row.append(row[0])                                   # duplicate column in a row of A
        ciA += row
        rpA.append(len(ciA))
        m = int(rng.integers(0, ncol + 1))
        cols = list(colnames[rng.choice(ncol, size=m, replace=False)])
        if m and rng.random() < 0.15:
            cols.append(cols[-1])                                # duplicate block in a row of X
        ciX += cols
        rpX.append(len(ciX))
    if not ciX:                                                  # X needs at least one block
        ciX, rpX = [int(colnames[0])], [0] + [1] * mb
    # B: the first block of every column of X, plus a random subset of further (row, column) positions
    seen_col, rpB, ciB = set(), [0], []
    for r in range(mb):
        done = set()
        for c in ciX[rpX[r]:rpX[r + 1]]:
            if c in done:
                continue
            if c not in seen_col or rng.random() < 0.3:
                ciB.append(c); seen_col.add(c); done.add(c)
        rpB.append(len(ciB))
    off = int(rng.integers(0, 2))
    z = lambda n: np.zeros((n, 4, 4), complex)
    return T.Problem(np.array(rpA) + off, np.array(ciA, dtype=np.int64) + off, z(len(ciA)), np.array(rpX) + off,
                     np.array(ciX, dtype=np.int64) + off, np.array(rpB) + off, np.array(ciB, dtype=np.int64) + off, z(len(ciB)),
                     None, 0.0, off)


def test_create_plan_random_patterns_bit_exact(oracle):
    """120 random ragged systems: the library's index analysis against the C restatement and, where it was built, against
    the reference itself -- pairs, starts, subset, colindx, original_bsrColIndX and the status, bit for bit"""
    rng = np.random.default_rng(20260)
    ref = oracle.Reference() if oracle.have_ref() else None
    nonempty = 0
    for case in range(120):
        pr = _random_pattern_problem(rng)
        an = oracle.analyse(pr)
        with T.Solver() as s:
            st = _create(pr, s.handle, s.plan)
            assert T.decode(st)[0] == T.decode(an["status"])[0], (case, st, an["status"])
            if st == 0:
                v = s.plan_view()
                for k in PLAN_KEYS + ("original_bsrColIndX",):
                    assert np.array_equal(v[k], an[k]), (case, k)
                assert v["nCols"] == an["nCols"]
                nonempty += (v["nPairs"] > 0)
        if ref is not None:
            r = ref.analyse(pr)
            assert T.decode(r["status"])[0] == T.decode(an["status"])[0], case
            if an["status"] == 0:
                for k in PLAN_KEYS + ("original_bsrColIndX",):
                    assert np.array_equal(r[k], an[k]), (case, k)
    assert nonempty > 60


def test_block_column_limit_and_stray_indices():
    """the compressed block column is 16 bit in the reference (colIndex_t, tfqmrgpu.hxx:59, asserted at
    tfqmrgpu_core.hxx:81): 65 536 block columns are accepted, one more is refused with NO_IMPLEMENTATION instead of being
    truncated; a stray column index (range 2^31) costs nothing -- the columns are compressed from the distinct values"""
    import tfqmrgpu_amd as T
    for ncols, want in ((65536, 0), (65537, 19)):
        cols = np.arange(ncols, dtype=np.int32)
        pr = T.Problem([0, 1], [0], np.eye(4)[None], [0, ncols], cols, [0, ncols], cols, np.zeros((ncols, 4, 4)), None, 1e-9)
        with T.Solver() as s:
            st = T.lib.tfqmrgpu_bsrsv_createPlan(s.handle, C.byref(s.plan), pr.mb, T._ptr(pr.rowPtrA), pr.nnzbA, T._ptr(pr.colIndA),
                                                 T._ptr(pr.rowPtrX), pr.nnzbX, T._ptr(pr.colIndX), T._ptr(pr.rowPtrB), pr.nnzbB,
                                                 T._ptr(pr.colIndB), 0, 0)
            assert T.decode(st)[0] == want
            if want == 0:
                v = s.plan_view()
                assert v["nCols"] == 65536 and v["colindx"][-1] == 65535
    cols = np.array([5, 2147483600, -2147483600], dtype=np.int32)
    pr = T.Problem([0, 1], [0], np.eye(4)[None], [0, 3], cols, [0, 3], cols, np.zeros((3, 4, 4)), None, 1e-9)
    with T.Solver() as s:
        s.create_plan(pr)
        v = s.plan_view()
        assert v["nCols"] == 3 and list(v["colindx"]) == [1, 2, 0] and list(v["original_bsrColIndX"]) == [-2147483600, 5, 2147483600]


def test_block_rows_beyond_the_reference_s_int_overflow(oracle):
    """the reference's createPlan checks `nnzbA > mb*mb` in 32-bit int (tfqmrgpu.cu:169): with 65 536 block rows the product
    wraps to 0 and every system is refused with status 14 -- BASELINE config 5 (512 x 512 = 262 144 block rows) cannot run on
    the reference.  This library has no such limit (documented deviation); bench.py's cpu_baseline falls
    back from the reference's CPU library to the oracle there."""
    import tfqmrgpu_amd as T
    mb = 65536
    rp = np.arange(mb + 1, dtype=np.int32)
    idx = np.arange(mb, dtype=np.int32)
    zero = np.zeros(mb, np.int32)
    bp = np.minimum(rp, 1).astype(np.int32)                      # one block of B, in row 0
    pr = T.Problem(rp, idx, np.broadcast_to(np.eye(4), (mb, 4, 4)), rp, zero, bp, [0], np.eye(4)[None], None, 1e-9)
    with T.Solver() as s:
        s.create_plan(pr)                                        # raises on any status but 0
        v = s.plan_view()
        assert v["nRows"] == mb and v["nCols"] == 1 and v["nPairs"] == mb
    if oracle.have_ref():
        st, h, plan = oracle.Reference().create_plan(pr)
        assert T.decode(st)[:2] == (14, 169)


def test_kept_profiler_figures_name_their_kernel():
    """profiles/pmc_traffic.json (scripts/pmc_to_traffic.py): every workload entry says which kernel family, revision and round its bytes were
    measured on -- bench.py drops `traffic` when the running plan's family differs (tfqmrgpuExt_getMultiplyKernel)"""
    import json
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    entries = {k: v for k, v in d.items() if not k.startswith("_")}
    assert entries
    for wl, e in entries.items():
        assert e.get("_kernel", "").startswith("k_spmm_") and e.get("_sha") and e.get("_round"), wl
        assert e["_kernels"].get("spmm_v4_dot", "").startswith(e["_kernel"] + "<"), wl
        assert all(isinstance(v, int) and v > 0 for k, v in e.items() if not k.startswith("_")), wl


def test_round4_extensions_refuse_what_they_cannot_do():
    """The additive calls of round 4 on the host side (no device call is reached): the kernel-family getter needs a registered buffer, the prepared
    launch order of the stand-alone multiply needs its lists, a null order is released as what it is."""
    pr = PR.stencil_2d(4, 3, 16, 16, 2, seed=3)
    s = T.Solver()
    s.create_plan(pr)
    s.buffer_size(16, 16, "z")
    buf = C.create_string_buffer(64)
    assert T.decode(T.lib.tfqmrgpuExt_getMultiplyKernel(s.plan, buf, 64))[0] == 7          # no buffer yet: TFQMRGPU_POINTER_INVALID
    assert T.decode(T.lib.tfqmrgpuExt_getMultiplyKernel(s.plan, None, 64))[0] == 7
    assert T.decode(T.lib.tfqmrgpuExt_getMultiplyKernel(None, buf, 64))[0] == 7
    order = C.c_void_p(None)
    assert T.decode(T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, b"z", 16, 16, 10, None, None, 4, C.byref(order)))[0] == 7
    assert T.decode(T.lib.tfqmrgpuExt_multiplyPrepare(s.handle, b"z", 16, 16, 10, None, None, 4, None))[0] == 7
    assert T.lib.tfqmrgpuExt_multiplyRelease(None) == 0
    assert T.decode(T.lib.tfqmrgpuExt_multiplyOrdered(s.handle, b"z", 16, 16, 10, None, None, None, None, None, None))[0] == 7
    s.close()
