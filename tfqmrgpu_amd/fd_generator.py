"""Finite-difference example problems A*X == B (block-sparse screened-Laplace operator, unit-block
right-hand sides inside a source cluster, X truncated to a target sphere around each source).

Own implementation of what the reference's generator produces
(real-space/tfQMRgpu example/tfqmrgpu_generate_FD_example.cxx:303-883; CLI
`rsb rtb block_edge dimension energy reference echo nFD`, defaults 1.75 6.75 2 3 0.0 n 5 4, :915-923):
same index lists, same integer stencil blocks, and -- through `to_xml` -- the same bytes in
FD_problem.xml (checked against the fixtures in tests/golden/).  Needed because BASELINE config 2
("generate_FD_example -> tfQMR", 16x16 complex<double> blocks) must be produced on the GPU box where
the reference does not exist.  Quirk kept on purpose: the energy shift is subtracted once per grid
point of a block, i.e. block_edge^dimension times (reference :475-481).
"""
import numpy as np

from . import Problem

FD_COEFF = {  # minus-Laplacian stencils, integer numerators over a common denominator (:352-388)
    1: (1, [2, -1]),
    4: (5040, [14350, -8064, 1008, -128, 9]),
    6: (831600, [2480478, -1425600, 222750, -44000, 7425, -864, 50]),
    8: (302702400, [924708642, -538137600, 94174080, -22830080, 5350800, -1053696, 156800, -15360, 735]),
}


def _cluster(center, radius, dim):
    """block coordinates inside `radius` (float32) of `center`, x fastest (reference create_cluster, :266-300)"""
    radius = np.float32(radius)
    irad = int(np.ceil(radius))
    rad2 = np.float32(radius * radius)
    ax = [np.arange(center[d] - irad, center[d] + irad + 1) if d < dim else np.array([0]) for d in range(3)]
    z, y, x = np.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    d2 = (center[0] - x) ** 2 + ((center[1] - y) ** 2 if dim > 1 else 0) + ((center[2] - z) ** 2 if dim > 2 else 0)
    keep = d2.astype(np.float32) <= rad2
    return np.stack([x[keep], y[keep] * (dim > 1), z[keep] * (dim > 2)], axis=1).astype(np.int64)


def _key(xyz):
    """the reference packs signed byte coordinates into one integer (little endian index4_t, :243-256)"""
    return (xyz[:, 0] & 0xFF) | ((xyz[:, 1] & 0xFF) << 8) | ((xyz[:, 2] & 0xFF) << 16)


class FDExample:
    def __init__(self, rsb=1.75, rtb=6.75, block_edge=2, dimension=3, energy=0.0, nFD=4, tolerance=1e-9):
        if nFD not in FD_COEFF:
            nFD = 1
        BE, D = int(block_edge), int(dimension)
        self.rsb, self.rtb = np.float32(abs(rsb)), np.float32(abs(rtb))
        self.BE, self.D, self.nFD, self.tolerance = BE, D, nFD, tolerance
        BS = self.BS = BE ** D
        denom, coeff = FD_COEFF[nFD]
        self.denom = denom

        # stencil blocks around the origin: centre, then +-1, +-2, ... along each direction (:403-430)
        sr = (nFD - 1) // BE + 1
        origin, seen = [], {}
        for isr in range(sr + 1):
            for ipm in (1, -1):
                for d in range(D):
                    xyz = [0, 0, 0]
                    xyz[d] = isr * ipm
                    if tuple(xyz) not in seen:
                        seen[tuple(xyz)] = len(origin)
                        origin.append(tuple(xyz))
        self.origin = origin
        sub = int(np.floor(abs(denom * energy) + 0.5) * np.sign(denom * energy))  # std::round
        self.energy_used = sub / float(denom)
        stencil = np.zeros((len(origin), BS, BS), dtype=np.int64)
        for z in range(BE if D > 2 else 1):
            for y in range(BE if D > 1 else 1):
                for x in range(BE):
                    p = (x, y, z)
                    ib = (z * BE + y) * BE + x
                    for d in range(D):
                        for iFD in range(-nFD, nFD + 1):
                            j = p[d] + iFD
                            shift, m = j // BE, j % BE
                            q = list(p)
                            q[d] = m
                            jb = (q[2] * BE + q[1]) * BE + q[0]
                            s = [0, 0, 0]
                            s[d] = shift
                            stencil[seen[tuple(s)], ib, jb] += coeff[abs(iFD)]
                    stencil[0][np.arange(BS), np.arange(BS)] -= sub   # once per grid point (reference quirk)
        self.stencil = stencil

        # sources, targets, rows (:484-540)
        src = _cluster((0, 0, 0), self.rsb / np.float32(BE), D)
        self.n_sources = len(src)
        tkeys, tsrc = [], []
        for isrc, c in enumerate(src):
            t = _cluster(tuple(int(v) for v in c), self.rtb / np.float32(BE), D)
            tkeys.append(_key(t))
            tsrc.append(np.full(len(t), isrc, dtype=np.int64))
        tkeys, tsrc = np.concatenate(tkeys), np.concatenate(tsrc)
        row_keys = np.unique(tkeys)                       # rows are numbered by ascending packed coordinate
        self.nrows = len(row_keys)
        trow = np.searchsorted(row_keys, tkeys)
        order = np.lexsort((tsrc, trow))                  # per row: ascending source index
        self.rowPtrX = np.concatenate([[0], np.cumsum(np.bincount(trow, minlength=self.nrows))]).astype(np.int32)
        self.colIndX = tsrc[order].astype(np.int32)

        srow = np.searchsorted(row_keys, _key(src))       # B: unit block at the row of each source (:577-600)
        border = np.argsort(srow, kind="stable")
        self.rowPtrB = np.concatenate([[0], np.cumsum(np.bincount(srow, minlength=self.nrows))]).astype(np.int32)
        self.colIndB = border.astype(np.int32)

        # A: for every row the stencil blocks that land on an existing row, in stencil order (:664-694)
        coords = np.stack([row_keys & 0xFF, (row_keys >> 8) & 0xFF, (row_keys >> 16) & 0xFF], axis=1)
        signed = np.where(coords > 127, coords - 256, coords)
        self.row_xyz = signed                             # block coordinates of every row (probes of the row order use them)
        ob = np.array(origin, dtype=np.int64)
        nk = ((signed[:, None, 0] + ob[None, :, 0]) & 0xFF) | (((signed[:, None, 1] + ob[None, :, 1]) & 0xFF) << 8) \
            | (((signed[:, None, 2] + ob[None, :, 2]) & 0xFF) << 16)
        pos = np.searchsorted(row_keys, nk)
        pos_c = np.minimum(pos, self.nrows - 1)
        hit = row_keys[pos_c] == nk
        self.rowPtrA = np.concatenate([[0], np.cumsum(hit.sum(axis=1))]).astype(np.int32)
        self.colIndA = pos_c[hit].astype(np.int32)
        self.indirA = np.broadcast_to(np.arange(len(origin)), hit.shape)[hit].astype(np.int32)

    # ---- as a solver problem (what the XML reader would return) -------------------------------------
    def problem(self):
        scale = float("%.16e" % (1.0 / self.denom))       # the value that travels through the XML file
        A = (self.stencil.astype(np.float64) * scale)[self.indirA].astype(np.complex128)
        B = np.broadcast_to(np.eye(self.BS, dtype=np.complex128), (len(self.colIndB), self.BS, self.BS)).copy()
        return Problem(self.rowPtrA, self.colIndA, A, self.rowPtrX, self.colIndX, self.rowPtrB, self.colIndB, B,
                       None, self.tolerance, 0)

    # ---- as FD_problem.xml (reference :853-878 and xml_export_*, :156-240) -------------------------
    def to_xml(self):
        out = ['<?xml version="1.0"?>\n<LinearProblem problem_kind="A*X==B"\n'
               '               generator_version="0.1" tolerance="%.3e">\n' % self.tolerance,
               "  <!-- input: radius_source_blocks=%g radius_target_blocks=%g\n\t\t block_edge=%d dimensions=%d"
               " energy=%g finite_difference=%d -->\n" % (float(self.rsb), float(self.rtb), self.BE, self.D,
                                                         self.energy_used, self.nFD)]

        def seq(values):
            return "".join(("\n" if (i & 0xF) == 0 else " ") + "%d" % v for i, v in enumerate(values))

        def operator(name, rowptr, colind, indirection, blocks, scale):
            o = ['  <BlockSparseMatrix id="%s">\n    <SparseMatrix type="CSR">\n      <CompressedSparseRow>\n' % name,
                 '        <NonzerosPerRow rows="%d">%s\n        </NonzerosPerRow>\n' % (len(rowptr) - 1, seq(np.diff(rowptr))),
                 '        <ColumnIndex nonzeros="%d">%s\n        </ColumnIndex>\n' % (len(colind), seq(colind)),
                 "      </CompressedSparseRow>\n"]
            if indirection is not None:
                o.append('      <Indirection nonzeros="%d">%s\n      </Indirection>\n' % (len(colind), seq(indirection)))
            o.append("    </SparseMatrix>\n")
            o.append('    <DataTensor type="real" rank="3" dimensions="%d %d %d"' % (len(blocks), self.BS, self.BS))
            if scale != 1:
                o.append(' scale="%.16e"' % scale)
            o.append(">\n")
            for b in blocks:
                for row in b:
                    o.append("".join("%.15g " % float(v) for v in row) + "\n")
                if self.BS > 1:
                    o.append("\n")
            o.append("    </DataTensor>\n  </BlockSparseMatrix>\n")
            return "".join(o)

        out.append(operator("A", self.rowPtrA, self.colIndA, self.indirA, self.stencil, 1.0 / self.denom))
        out.append(operator("B", self.rowPtrB, self.colIndB, np.zeros(len(self.colIndB), int), [np.eye(self.BS, dtype=int)], 1))
        out.append(operator("X", self.rowPtrX, self.colIndX, None, [], 1))
        out.append("</LinearProblem>\n")
        return "".join(out)


def fd_problem(rsb=1.75, rtb=6.75, block_edge=2, dimension=3, energy=0.0, nFD=4, tolerance=1e-9):
    return FDExample(rsb, rtb, block_edge, dimension, energy, nFD, tolerance).problem()


def main(argv=None):
    """same positional arguments as the reference generator; writes FD_problem.xml"""
    import sys
    a = list(sys.argv[1:] if argv is None else argv)
    rsb = float(a[0]) if len(a) > 0 else 1.75
    rtb = float(a[1]) if len(a) > 1 else 6.75
    be = int(a[2]) if len(a) > 2 else 2
    dim = int(a[3]) if len(a) > 3 else 3
    energy = float(a[4]) if len(a) > 4 else 0.0
    nfd = int(a[7]) if len(a) > 7 else 4
    ex = FDExample(rsb, rtb, be, dim, energy, nfd)
    with open("FD_problem.xml", "w") as f:
        f.write(ex.to_xml())
    print("# FD_problem.xml: %d rows, nnzb A=%d X=%d B=%d" % (ex.nrows, len(ex.colIndA), len(ex.colIndX), len(ex.colIndB)))


if __name__ == "__main__":
    main()
