"""TEST INFRASTRUCTURE -- ctypes access to the CPU oracle (oracle/liboracle.so, this repo's plain-C
restatement) and, where it was built, to the reference itself (oracle/_ref/, compiled from
/root/reference by oracle/Makefile: buildable only where the reference's sources are, i.e. in the build container; the
BUILT files are git-ignored but travel with gpurun snapshots, so that bench.py's cpu_baseline leg can time the reference's own
CPU library on the GPU box's host cores -- DESIGN.md section 5).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libtfQMRgpu_ref.so")
REF_PEEK_SO = os.path.join(HERE, "_ref", "libref_peek.so")
REF_FDGEN = os.path.join(HERE, "_ref", "generate_FD_example")

REDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build()
        _lib = C.CDLL(ORACLE_SO)
        _lib.tfqo_shadow_glibc.argtypes = [C.c_void_p, C.c_uint64]
        _lib.tfqo_shadow_glibc.restype = None
        # a GPU box shows far more hardware threads than its CPU share: never let OpenMP grab them all
        set_threads(min(8, os.cpu_count() or 1))
    return _lib


def set_threads(n):
    return _lib.tfqo_set_threads(C.c_int(n))


def have_ref():
    return os.path.exists(REF_SO) and os.path.exists(REF_PEEK_SO)


# ---- helpers: complex blocks <-> native planes -----------------------------------------------------
def to_native(blocks, real):
    """complex [n, R, C] -> real [n, 2, R, C] (the device-native RRRRIIII order)"""
    b = np.asarray(blocks)
    return np.ascontiguousarray(np.stack([b.real, b.imag], axis=1), dtype=real)


def from_native(planes):
    return planes[:, 0] + 1j * planes[:, 1]


def a_native(A, real, transA="n"):
    """native A[n][2][k][i] = op(A)[i][k]  (stored transposed, tfqmrgpu.cu:514-517)"""
    A = np.asarray(A)
    t = transA.lower()
    M = {"n": A, "t": A.transpose(0, 2, 1), "*": A.conj(), "c": A.conj().transpose(0, 2, 1), "h": A.conj().transpose(0, 2, 1)}[t]
    return to_native(M.transpose(0, 2, 1), real)


def shadow_glibc(n):
    v = np.zeros(n, dtype=np.float32)
    lib().tfqo_shadow_glibc(_p(v), n)
    return v


# ---- integer analysis ------------------------------------------------------------------------------
def analyse(pr):
    L = lib()
    cap = max(1, pr.nnzbX) * max(1, int(np.max(np.diff(pr.rowPtrA))) if pr.nnzbA else 1)
    starts = np.zeros(pr.nnzbX + 1, np.uint32)
    pairs = np.zeros(2 * cap, np.uint32)
    subset = np.zeros(max(1, pr.nnzbB), np.uint32)
    colindx = np.zeros(max(1, pr.nnzbX), np.uint16)
    orig = np.zeros(max(1, pr.nnzbX), np.int32)
    nPairs, nCols = C.c_uint64(0), C.c_uint32(0)
    L.tfqo_analyse.restype = C.c_int32
    st = L.tfqo_analyse(C.c_int32(pr.mb), _p(pr.rowPtrA), C.c_int32(pr.nnzbA), _p(pr.colIndA),
                        _p(pr.rowPtrX), C.c_int32(pr.nnzbX), _p(pr.colIndX),
                        _p(pr.rowPtrB), C.c_int32(pr.nnzbB), _p(pr.colIndB),
                        C.c_int32(pr.index_offset), C.c_uint64(cap),
                        _p(starts), _p(pairs), C.byref(nPairs), _p(subset), _p(colindx), _p(orig), C.byref(nCols))
    n = int(nPairs.value)
    return dict(status=st, nPairs=n, nCols=int(nCols.value), starts=starts, pairs=pairs[:2 * n].copy(),
                subset=subset[:pr.nnzbB].copy(), colindx=colindx[:pr.nnzbX].copy(), original_bsrColIndX=orig[:nCols.value].copy())


# ---- floating point --------------------------------------------------------------------------------
def spmm(precision, LM, LN, starts, pairs, A_nat, X_nat):
    L = lib()
    Y = np.zeros_like(X_nat)
    fn = L.tfqo_spmm_z if precision == "z" else L.tfqo_spmm_c
    fn.restype = None
    fn(C.c_int(LM), C.c_int(LN), C.c_uint32(len(starts) - 1), _p(starts), _p(pairs), _p(A_nat), _p(X_nat), _p(Y))
    return Y


def solve(pr, precision="z", threshold=None, max_iterations=2000, transA="n", v3=None, plan=None, reduce=None, dump_iteration=None):
    """tfQMR on the CPU with the documented (CUDA-path) semantics of the reference.
    v3: float array [nnzbX*2*LM*LN] (default: the glibc rand() sequence of the reference CPU path).
    dump_iteration: info["vectors"] = {1: x, 4: v4, ... 9: v9} (complex [nnzbX][LM][LN]) at the end of that iteration,
    before its stopping test (tfqmrgpu_core.hxx:233)."""
    L = lib()
    real = np.float64 if precision == "z" else np.float32
    an = plan or analyse(pr)
    assert an["status"] == 0, an["status"]
    A = a_native(pr.A, real, transA)
    B = to_native(pr.B, real)
    X = np.zeros((pr.nnzbX, 2, pr.LM, pr.LN), dtype=real)
    if v3 is None:
        v3 = shadow_glibc(pr.nnzbX * 2 * pr.LM * pr.LN)
    v3 = np.ascontiguousarray(v3, dtype=np.float32)
    it, res, flops, nh = C.c_int32(0), C.c_double(0), C.c_double(0), C.c_int32(0)
    hist = np.zeros(max(1, max_iterations), np.float64)
    cb = REDUCE_CB(reduce) if reduce else C.cast(None, REDUCE_CB)
    fn = L.tfqo_solve_z if precision == "z" else L.tfqo_solve_c
    fn.restype = C.c_int
    hook = L.tfqo_set_dump_z if precision == "z" else L.tfqo_set_dump_c
    hook.restype = None
    dump = np.zeros((7,) + X.shape, dtype=real) if dump_iteration else None
    if dump is not None:
        hook(C.c_int(dump_iteration), _p(dump))
    st = fn(C.c_int(pr.LM), C.c_int(pr.LN), C.c_uint32(pr.nnzbX), C.c_uint32(pr.nnzbB), C.c_uint32(an["nCols"]),
            _p(an["starts"]), _p(an["pairs"]), _p(an["subset"]), _p(an["colindx"]),
            _p(A), _p(B), _p(v3), _p(X),
            C.c_double(pr.tolerance if threshold is None else threshold), C.c_int(max_iterations),
            C.byref(it), C.byref(res), C.byref(flops), _p(hist), C.byref(nh), cb, None)
    info = dict(iterations=it.value, residual=res.value, flops=flops.value, bound_history=hist[:nh.value].copy())
    if dump is not None:
        hook(C.c_int(0), None)
        info["vectors"] = {w: from_native(dump[n]).astype(np.complex128) for n, w in enumerate((1, 4, 5, 6, 7, 8, 9))}
    return st, from_native(X).astype(np.complex128), info


# ---- the reference itself (container only) ---------------------------------------------------------
_FLIP = {"n": "t", "t": "n", "c": "*", "h": "*", "*": "c"}


class Reference:
    """The compiled reference CPU library.  Its CPU multiply uses A un-transposed where its CUDA
    path and manual use the transpose (tfqmrgpu_blocksparse.hxx:167-168 vs tfqmrgpu_blockmult.hxx:54),
    so `solve` passes A's flag flipped to obtain the documented semantics."""

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        self.peek = C.CDLL(REF_PEEK_SO)
        self.libc = C.CDLL("libc.so.6")
        self.peek.refpeek_sizes.restype = None
        self.peek.refpeek_copy.restype = None

    def create_plan(self, pr, echo=0):
        h, plan = C.c_void_p(None), C.c_void_p(None)
        self.lib.tfqmrgpuCreateHandle(C.byref(h))
        st = self.lib.tfqmrgpu_bsrsv_createPlan(h, C.byref(plan), C.c_int(pr.mb),
                                                _p(pr.rowPtrA), C.c_int(pr.nnzbA), _p(pr.colIndA),
                                                _p(pr.rowPtrX), C.c_int(pr.nnzbX), _p(pr.colIndX),
                                                _p(pr.rowPtrB), C.c_int(pr.nnzbB), _p(pr.colIndB),
                                                C.c_int(pr.index_offset), C.c_int(echo))
        return st, h, plan

    def analyse(self, pr):
        st, h, plan = self.create_plan(pr)
        if st != 0:
            return dict(status=st)
        nP, nC, nX, nB = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self.peek.refpeek_sizes(plan, C.byref(nP), C.byref(nC), C.byref(nX), C.byref(nB))
        pairs = np.zeros(max(1, 2 * nP.value), np.uint32)
        starts = np.zeros(nX.value + 1, np.uint32)
        subset = np.zeros(max(1, nB.value), np.uint32)
        colindx = np.zeros(max(1, nX.value), np.uint16)
        orig = np.zeros(max(1, nC.value), np.int32)
        self.peek.refpeek_copy(plan, _p(pairs), _p(starts), _p(subset), _p(colindx), _p(orig))
        out = dict(status=0, nPairs=int(nP.value), nCols=int(nC.value), starts=starts, pairs=pairs[:2 * nP.value].copy(),
                   subset=subset[:nB.value].copy(), colindx=colindx[:nX.value].copy(), original_bsrColIndX=orig[:nC.value].copy())
        sizes = {}
        for prec in "zc":
            n = C.c_size_t(0)
            # LM/LN are only needed for the buffer size; the caller fills them in
            sizes[prec] = n
        self.lib.tfqmrgpu_bsrsv_destroyPlan(h, plan)
        self.lib.tfqmrgpuDestroyHandle(h)
        return out

    def solve(self, pr, precision="z", threshold=None, max_iterations=2000, transA="n"):
        """documented semantics (A flag flipped for the CPU library); v3 = rand() sequence from seed 1"""
        self.libc.srand(1)  # same sequence as a never-seeded process
        real = np.float64 if precision == "z" else np.float32
        cplx = np.complex128 if precision == "z" else np.complex64
        A = np.ascontiguousarray(pr.A.astype(cplx))
        B = np.ascontiguousarray(pr.B.astype(cplx))
        X = np.zeros((pr.nnzbX, pr.LM, pr.LN), dtype=cplx)
        it = C.c_int32(max_iterations)
        res = C.c_float(pr.tolerance if threshold is None else threshold)
        fn = self.lib.tfqmrgpu_bsrsv_z if precision == "z" else self.lib.tfqmrgpu_bsrsv_c
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        import sys
        sys.stdout.flush()
        os.dup2(devnull, 1)  # the reference prints unconditionally (tfqmrgpu_linalg.hxx:18-25)
        try:
            st = fn(C.c_int(pr.mb), C.c_int(pr.LM), C.c_int(pr.LN),
                    _p(pr.rowPtrA), C.c_int(pr.nnzbA), _p(pr.colIndA), _p(A), C.c_char(_FLIP[transA.lower()].encode()),
                    _p(pr.rowPtrX), C.c_int(pr.nnzbX), _p(pr.colIndX), _p(X), C.c_char(b"n"),
                    _p(pr.rowPtrB), C.c_int(pr.nnzbB), _p(pr.colIndB), _p(B), C.c_char(b"n"),
                    C.byref(it), C.byref(res), C.c_int(pr.index_offset), C.c_int(0))
            self.libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        return st, X.astype(np.complex128), dict(iterations=it.value, residual=float(res.value))

    def solve_staged(self, pr, precision="z", threshold=None, max_iterations=2000, transA="n"):
        """staged API, gives double-precision residual + flops (getInfo)"""
        self.libc.srand(1)
        L = self.lib
        cplx = np.complex128 if precision == "z" else np.complex64
        A = np.ascontiguousarray(pr.A.astype(cplx))
        B = np.ascontiguousarray(pr.B.astype(cplx))
        X = np.zeros((pr.nnzbX, pr.LM, pr.LN), dtype=cplx)
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(1)
        import sys
        sys.stdout.flush()
        os.dup2(devnull, 1)
        try:
            st, h, plan = self.create_plan(pr)
            assert st == 0, st
            L.tfqmrgpuSetStream(h, C.c_void_p(0))
            nbytes = C.c_size_t(0)
            L.tfqmrgpu_bsrsv_bufferSize(h, plan, C.c_int(pr.LM), C.c_int(pr.LM), C.c_int(pr.LN), C.c_int(pr.LN),
                                        C.c_char(precision.encode()), C.byref(nbytes))
            buf = C.c_void_p(None)
            L.tfqmrgpuCreateWorkspace(C.byref(buf), nbytes, C.c_char(b"d"))
            L.tfqmrgpu_bsrsv_setBuffer(h, plan, buf)
            pc = C.c_char(precision.encode())
            L.tfqmrgpu_bsrsv_setMatrix(h, plan, C.c_char(b"A"), _p(A), pc, C.c_int(pr.LM), C.c_int(pr.LM),
                                       C.c_char(_FLIP[transA.lower()].encode()), C.c_int(0x55))
            L.tfqmrgpu_bsrsv_setMatrix(h, plan, C.c_char(b"B"), _p(B), pc, C.c_int(pr.LN), C.c_int(pr.LM), C.c_char(b"n"), C.c_int(0x55))
            st = L.tfqmrgpu_bsrsv_solve(h, plan, C.c_double(pr.tolerance if threshold is None else threshold), C.c_int(max_iterations))
            res, fl, it = C.c_double(0), C.c_double(0), C.c_int32(0)
            L.tfqmrgpu_bsrsv_getInfo(h, plan, C.byref(res), C.byref(it), C.byref(fl), None)
            L.tfqmrgpu_bsrsv_getMatrix(h, plan, C.c_char(b"X"), _p(X), pc, C.c_int(pr.LN), C.c_int(pr.LM), C.c_char(b"n"), C.c_int(0x55))
            self.libc.fflush(None)
        finally:
            os.dup2(saved, 1)
            os.close(devnull)
            os.close(saved)
        return st, X.astype(np.complex128), dict(iterations=it.value, residual=res.value, flops=fl.value, buffer_bytes=nbytes.value)


def fd_xml(args, workdir):
    """run the reference's FD generator (container only); returns the path of FD_problem.xml"""
    os.makedirs(workdir, exist_ok=True)
    subprocess.check_call([REF_FDGEN] + [str(a) for a in args], cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return os.path.join(workdir, "FD_problem.xml")
