"""Problem sources for tests and benchmarks: the `<LinearProblem>` XML reader and synthetic
block-sparse systems.  Host-side plumbing (numpy); the solver itself is the HIP library.

read_xml follows the schema the reference reads in
real-space/tfQMRgpu tfQMRgpu/include/tfqmrgpu_example_xml_reader.hxx:125-292:
LinearProblem@tolerance -> 3 x BlockSparseMatrix@id{A,B,X} -> SparseMatrix/CompressedSparseRow/
{NonzerosPerRow@rows | RowStart@rows}, ColumnIndex@nonzeros, optional Indirection,
DataTensor@type{real,complex}@rank@dimensions@scale.  Values = data[indirection[block]] * scale,
real tensors get a zero imaginary part.
"""
import xml.etree.ElementTree as ET

import numpy as np

from . import Problem


def _seq(text, dtype):
    return np.array(text.split(), dtype=dtype) if text and text.strip() else np.zeros(0, dtype)


def read_xml(path):
    root = ET.parse(path).getroot()
    if root.tag != "LinearProblem":
        raise ValueError("%s: no <LinearProblem> root" % path)
    tol = float(root.attrib.get("tolerance", "0"))
    ops = {}
    for bsm in root.findall("BlockSparseMatrix"):
        name = bsm.attrib.get("id", "?")[0]
        sm = bsm.find("SparseMatrix")
        csr = sm.find("CompressedSparseRow")
        nzpr = csr.find("NonzerosPerRow")
        if nzpr is not None:
            per_row = _seq(nzpr.text, np.int64)
            rowptr = np.concatenate([[0], np.cumsum(per_row)]).astype(np.int32)
        else:
            rowptr = _seq(csr.find("RowStart").text, np.int64).astype(np.int32)
        colind = _seq(csr.find("ColumnIndex").text, np.int64).astype(np.int32)
        nnzb = len(colind)
        ind = sm.find("Indirection")
        indirection = _seq(ind.text, np.int64) if ind is not None else np.arange(nnzb)
        dt = bsm.find("DataTensor")
        blocks = None
        if dt is not None:
            scale = float(dt.attrib.get("scale", "1"))
            is_complex = dt.attrib.get("type", "complex")[0].lower() == "c"
            dims = [int(v) for v in dt.attrib.get("dimensions", "0 0 0").split()]
            data = _seq(dt.text, np.float64)
            if dims[0] > 0:
                if is_complex:
                    data = data.reshape(dims[0], dims[1], dims[2], 2)
                    src = data[..., 0] + 1j * data[..., 1]
                else:
                    src = data.reshape(dims[0], dims[1], dims[2]).astype(np.complex128)
                blocks = src[indirection] * scale
            else:
                blocks = np.zeros((nnzb, dims[1], dims[2]), dtype=np.complex128)
        ops[name] = (rowptr, colind, blocks)
    (rA, cA, A), (rB, cB, B), (rX, cX, X) = ops["A"], ops["B"], ops["X"]
    return Problem(rA, cA, A, rX, cX, rB, cB, B, X, tol, 0)


# ---- deterministic pseudo-random values -----------------------------------------------------------
def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    return x ^ (x >> np.uint64(31))


def hashed_uniform(seed, shape):
    """uniform [-1, 1) from a counter-based hash: the same values for any generation order"""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        h = _splitmix64(np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0xD1342543DE82EF95))
    return ((h >> np.uint64(11)).astype(np.float64) * (2.0 ** -52) - 1.0).reshape(shape)


# ---- synthetic systems -------------------------------------------------------------------------------
def julia_kat(ldA=4, ldB=5, mb=7):
    """Known-answer test of the reference's Julia example (example/tfqmrgpu_Julia_example.jl:41-66):
    1-D finite-difference Laplacian (2,-1) (x) I, B = e_last with per-RHS phases i^p;
    the solution is the straight line k/(mb+1)."""
    rpA, ciA, blocks = [0], [], []
    for ib in range(mb):
        for jb in range(max(0, ib - 1), min(mb - 1, ib + 1) + 1):
            ciA.append(jb)
            blocks.append((2.0 if ib == jb else -1.0) * np.eye(ldA))
        rpA.append(len(ciA))
    A = np.array(blocks, dtype=np.complex128)
    rpX, ciX = np.arange(mb + 1), np.zeros(mb, dtype=np.int32)
    rpB = np.zeros(mb + 1, dtype=np.int32)
    rpB[mb] = 1
    B = np.zeros((1, ldA, ldB), dtype=np.complex128)
    for i in range(ldB):
        B[0, i % ldA, i] = 1j ** (i // ldA)
    return Problem(rpA, ciA, A, rpX, ciX, rpB, [0], B, None, 1.2e-8)


def dense_random(mb=5, LM=4, LN=4, ncols=2, seed=7, shift=8.0):
    """All blocks present, no symmetry anywhere: catches a transposed A or a swapped
    Re/Im plane, which symmetric stencils hide."""
    A = hashed_uniform(seed, (mb * mb, LM, LM)) + 1j * hashed_uniform(seed + 1, (mb * mb, LM, LM))
    for i in range(mb):
        A[i * mb + i] += shift * np.eye(LM)
    rpA = np.arange(mb + 1) * mb
    ciA = np.tile(np.arange(mb), mb)
    rpX = np.arange(mb + 1) * ncols
    ciX = np.tile(np.arange(ncols), mb)
    B = hashed_uniform(seed + 2, (mb * ncols, LM, LN)) + 1j * hashed_uniform(seed + 3, (mb * ncols, LM, LN))
    return Problem(rpA, ciA, A, rpX, ciX, rpX, ciX, B, None, 1e-10)


def stencil_2d(nx, ny, LM, LN, ncols, seed=1, radius=None, points=5):
    """Block 5-point (or 9/13-point) stencil on an nx x ny grid with random blocks, made block
    diagonally dominant; X dense in `ncols` block columns (or inside `radius` grid steps of the
    source of each column); B = one random block per column.  Used for BASELINE configs 3-5."""
    mb = nx * ny
    offs = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)]
    if points >= 9:
        offs += [(2, 0), (-2, 0), (0, 2), (0, -2)]
    if points >= 13:
        offs += [(1, 1), (1, -1), (-1, 1), (-1, -1)]
    rpA, ciA = [0], []
    for y in range(ny):
        for x in range(nx):
            for dx, dy in offs:
                xx, yy = x + dx, y + dy
                if 0 <= xx < nx and 0 <= yy < ny:
                    ciA.append(yy * nx + xx)
            rpA.append(len(ciA))
    ciA = np.array(ciA, dtype=np.int32)
    nnzbA = len(ciA)
    A = (hashed_uniform(seed, (nnzbA, LM, LM)) + 1j * hashed_uniform(seed + 11, (nnzbA, LM, LM))) / (LM * len(offs))
    rows = np.repeat(np.arange(mb), np.diff(rpA))
    diag = np.nonzero(rows == ciA)[0]
    A[diag] += 2.0 * np.eye(LM)
    src = [(c * mb) // ncols + (mb // ncols) // 2 for c in range(ncols)]  # source row of each column
    rpX, ciX = [0], []
    for r in range(mb):
        for c in range(ncols):
            if radius is None:
                ciX.append(c)
            else:
                sx, sy = src[c] % nx, src[c] // nx
                if (r % nx - sx) ** 2 + (r // nx - sy) ** 2 <= radius * radius:
                    ciX.append(c)
        rpX.append(len(ciX))
    rpB, ciB = [0], []
    for r in range(mb):
        for c in range(ncols):
            if src[c] == r:
                ciB.append(c)
        rpB.append(len(ciB))
    B = hashed_uniform(seed + 5, (len(ciB), LM, LN)) + 1j * hashed_uniform(seed + 6, (len(ciB), LM, LN))
    return Problem(rpA, ciA, A, rpX, ciX, rpB, ciB, B, None, 1e-9)


def dense_reference_solution(pr):
    """Solve the pattern-truncated system column by column with dense LAPACK (numpy): for a block
    column the unknowns are its X blocks, rows/columns of A outside the pattern are dropped
    (that is what the pair list encodes)."""
    off = pr.index_offset
    rowsX = np.repeat(np.arange(pr.mb), np.diff(pr.rowPtrX))
    rowsA = np.repeat(np.arange(pr.mb), np.diff(pr.rowPtrA))
    rowsB = np.repeat(np.arange(pr.mb), np.diff(pr.rowPtrB))
    X = np.zeros((pr.nnzbX, pr.LM, pr.LN), dtype=np.complex128)
    for col in np.unique(pr.colIndX):
        xs = np.nonzero(pr.colIndX == col)[0]
        pos = {int(rowsX[q]): n for n, q in enumerate(xs)}
        n = len(xs)
        M = np.zeros((n * pr.LM, n * pr.LM), dtype=np.complex128)
        for q in range(pr.nnzbA):
            r, k = int(rowsA[q]), int(pr.colIndA[q] - off)
            if r in pos and k in pos:
                M[pos[r] * pr.LM:(pos[r] + 1) * pr.LM, pos[k] * pr.LM:(pos[k] + 1) * pr.LM] += pr.A[q]
        rhs = np.zeros((n * pr.LM, pr.LN), dtype=np.complex128)
        for q in np.nonzero(pr.colIndB == col)[0]:
            r = int(rowsB[q])
            rhs[pos[r] * pr.LM:(pos[r] + 1) * pr.LM] += pr.B[q]
        sol = np.linalg.solve(M, rhs)
        for q in xs:
            X[q] = sol[pos[int(rowsX[q])] * pr.LM:(pos[int(rowsX[q])] + 1) * pr.LM]
    return X
