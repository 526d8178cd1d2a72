"""Python binding of the MI355X-native libtfQMRgpu.so (C-ABI in include/tfqmrgpu.h).

This is plumbing for tests and bench.py, not a second implementation: every call goes through
ctypes into the HIP library.  There is no CPU fallback -- if the shared library is missing the
import fails loudly (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C tfqmrgpu_amd/csrc``).

The function names and argument order mirror the reference interface
(real-space/tfQMRgpu tfQMRgpu/include/tfqmrgpu.h:16-156) the way its own ctypes example binds it
(example/tfqmrgpu_python_example.py:40-67).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFQMRGPU_LIB", os.path.join(_HERE, "lib", "libtfQMRgpu.so"))   # override for A/B builds
LAB_LIB_PATH = os.path.join(_HERE, "lib", "libtfQMRgpu_lab.so")   # the same sources with -DTFQ_LAB: reads the TFQMRGPU_* tuning switches

LAYOUT_RRRRIIII, LAYOUT_RRIIRRII, LAYOUT_RIRIRIRI = 0x0F, 0x33, 0x55
SHADOW_HASH, SHADOW_GLIBC_RAND = 0, 1
CODE_LINE, CODE_CHAR = 1000, 10000 * 1000

EXPORTED_SYMBOLS = [  # include/tfqmrgpu.h
    "tfqmrgpuPrintError", "tfqmrgpuGetErrorString", "tfqmrgpuCreateHandle", "tfqmrgpuDestroyHandle",
    "tfqmrgpuSetStream", "tfqmrgpuGetStream", "tfqmrgpuCreateWorkspace", "tfqmrgpuDestroyWorkspace",
    "tfqmrgpu_bsrsv_allowedBlockSizes", "tfqmrgpu_bsrsv_blockSizeMissing", "tfqmrgpu_bsrsv_createPlan",
    "tfqmrgpu_bsrsv_destroyPlan", "tfqmrgpu_bsrsv_bufferSize", "tfqmrgpu_bsrsv_setBuffer",
    "tfqmrgpu_bsrsv_getBuffer", "tfqmrgpu_bsrsv_setMatrix", "tfqmrgpu_bsrsv_getMatrix",
    "tfqmrgpu_bsrsv_solve", "tfqmrgpu_bsrsv_getInfo", "tfqmrgpu_bsrsv_z", "tfqmrgpu_bsrsv_c",
]
EXT_SYMBOLS = [  # include/tfqmrgpu_ext.h
    "tfqmrgpuExt_planView", "tfqmrgpuExt_getBoundHistory", "tfqmrgpuExt_setProfiling", "tfqmrgpuExt_getProfile",
    "tfqmrgpuExt_getProfileGated", "tfqmrgpuExt_getProfileFirst", "tfqmrgpuExt_getMultiplyKernel",
    "tfqmrgpuExt_setShadowMode",
    "tfqmrgpuExt_setShadowVector", "tfqmrgpuExt_getShadowVector", "tfqmrgpuExt_getWorkVector", "tfqmrgpuExt_multiply", "tfqmrgpuExt_multiplyPrepare", "tfqmrgpuExt_multiplyOrdered", "tfqmrgpuExt_multiplyRelease", "tfqmrgpuExt_applyOperator", "tfqmrgpuExt_shardColumns",
    "tfqmrgpuExt_freeShard", "tfqmrgpuExt_commUniqueId", "tfqmrgpuExt_commInit",
    "tfqmrgpuExt_commDestroy", "tfqmrgpuExt_setReduceCallback", "tfqmrgpuExt_setOperator",
    "tfqmrgpuExt_getRefinementHistory", "tfqmrgpuExt_setThreeProductMultiply",
]
FORTRAN_SYMBOLS = [  # tfqmrgpu_amd/csrc/tfq_fortran.c
    "tfqmrgpuprinterror_", "tfqmrgpucreatehandle_", "tfqmrgpudestroyhandle_", "tfqmrgpusetstream_",
    "tfqmrgpugetstream_", "tfqmrgpu_bsrsv_createplan_", "tfqmrgpu_bsrsv_destroyplan_",
    "tfqmrgpu_bsrsv_buffersize_", "tfqmrgpucreateworkspace_", "tfqmrgpudestroyworkspace_",
    "tfqmrgpu_bsrsv_setbuffer_", "tfqmrgpu_bsrsv_getbuffer_", "tfqmrgpu_bsrsv_setmatrix_c_",
    "tfqmrgpu_bsrsv_setmatrix_z_", "tfqmrgpu_bsrsv_getmatrix_c_", "tfqmrgpu_bsrsv_getmatrix_z_",
    "tfqmrgpu_bsrsv_solve_", "tfqmrgpu_bsrsv_getinfo_",
]


class PlanView(C.Structure):
    _fields_ = [("nRows", C.c_uint32), ("nCols", C.c_uint32), ("nnzbA", C.c_uint32), ("nnzbX", C.c_uint32),
                ("nnzbB", C.c_uint32), ("nPairs", C.c_uint64), ("pairs", C.POINTER(C.c_uint32)),
                ("starts", C.POINTER(C.c_uint32)), ("subset", C.POINTER(C.c_uint32)),
                ("colindx", C.POINTER(C.c_uint16)), ("original_bsrColIndX", C.POINTER(C.c_int32)),
                ("LM", C.c_int32), ("LN", C.c_int32), ("precision", C.c_char)]


class Shard(C.Structure):
    _fields_ = [("mb", C.c_int32), ("nnzbX", C.c_int32), ("nnzbB", C.c_int32),
                ("rowPtrX", C.POINTER(C.c_int32)), ("colIndX", C.POINTER(C.c_int32)),
                ("rowPtrB", C.POINTER(C.c_int32)), ("colIndB", C.POINTER(C.c_int32)),
                ("xBlocks", C.POINTER(C.c_int32)), ("bBlocks", C.POINTER(C.c_int32)),
                ("firstCol", C.c_int32), ("nCols", C.c_int32)]


REDUCE_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_int)
# tfqmrgpuOperator_t: (ctx, Y_d, X_d, colindx_d, nnzbX, nCols, lm, ln, precision, stream, *flops) -> status
OPERATOR_CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                          C.c_int, C.c_int, C.c_char, C.c_void_p, C.POINTER(C.c_double))


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise ImportError(
            "tfqmrgpu_amd: %s is missing -- the HIP library has not been built; there is no CPU fallback. "
            "Run __graft_entry__.build() or `make -C tfqmrgpu_amd/csrc`." % path)
    # PyTorch ships its own libamdhip64.so.7 (same soname as /opt/rocm's).  A process must run ONE HIP
    # runtime: when torch is going to be used (bench.py, tests: device memory, streams, torch.distributed)
    # it has to be loaded first, then this library binds to the runtime that is already there.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    I, P = C.c_int, C.c_void_p
    lib.tfqmrgpuGetErrorString.restype = C.c_char_p
    lib.tfqmrgpuGetErrorString.argtypes = [C.c_int32]
    lib.tfqmrgpuCreateHandle.argtypes = [C.POINTER(P)]
    lib.tfqmrgpuDestroyHandle.argtypes = [P]
    lib.tfqmrgpuSetStream.argtypes = [P, P]
    lib.tfqmrgpuGetStream.argtypes = [P, C.POINTER(P)]
    lib.tfqmrgpuCreateWorkspace.argtypes = [C.POINTER(P), C.c_size_t, C.c_char]
    lib.tfqmrgpuDestroyWorkspace.argtypes = [P]
    lib.tfqmrgpu_bsrsv_allowedBlockSizes.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), I]
    lib.tfqmrgpu_bsrsv_blockSizeMissing.argtypes = [I, I]
    lib.tfqmrgpu_bsrsv_createPlan.argtypes = [P, C.POINTER(P), I, P, I, P, P, I, P, P, I, P, I, I]
    lib.tfqmrgpu_bsrsv_destroyPlan.argtypes = [P, P]
    lib.tfqmrgpu_bsrsv_bufferSize.argtypes = [P, P, I, I, I, I, C.c_char, C.POINTER(C.c_size_t)]
    lib.tfqmrgpu_bsrsv_setBuffer.argtypes = [P, P, P]
    lib.tfqmrgpu_bsrsv_getBuffer.argtypes = [P, P, C.POINTER(P)]
    lib.tfqmrgpu_bsrsv_setMatrix.argtypes = [P, P, C.c_char, P, C.c_char, I, I, C.c_char, I]
    lib.tfqmrgpu_bsrsv_getMatrix.argtypes = [P, P, C.c_char, P, C.c_char, I, I, C.c_char, I]
    lib.tfqmrgpu_bsrsv_solve.argtypes = [P, P, C.c_double, I]
    lib.tfqmrgpu_bsrsv_getInfo.argtypes = [P, P, C.POINTER(C.c_double), C.POINTER(C.c_int32),
                                           C.POINTER(C.c_double), C.POINTER(C.c_double)]
    onecall = [I, I, I, P, I, P, P, C.c_char, P, I, P, P, C.c_char, P, I, P, P, C.c_char,
               C.POINTER(C.c_int32), C.POINTER(C.c_float), I, I]
    lib.tfqmrgpu_bsrsv_z.argtypes = onecall
    lib.tfqmrgpu_bsrsv_c.argtypes = onecall
    lib.tfqmrgpuExt_planView.argtypes = [P, C.POINTER(PlanView)]
    lib.tfqmrgpuExt_getBoundHistory.argtypes = [P, P, C.c_int32]
    lib.tfqmrgpuExt_setProfiling.argtypes = [P, I]
    lib.tfqmrgpuExt_getProfile.argtypes = [P, P, P]
    lib.tfqmrgpuExt_getProfileGated.argtypes = [P, P, P]
    lib.tfqmrgpuExt_getProfileFirst.argtypes = [P, P, P]
    lib.tfqmrgpuExt_getMultiplyKernel.argtypes = [P, P, C.c_int32]
    lib.tfqmrgpuExt_setShadowMode.argtypes = [P, I]
    lib.tfqmrgpuExt_setShadowVector.argtypes = [P, P, P]
    lib.tfqmrgpuExt_getShadowVector.argtypes = [P, P, P]
    lib.tfqmrgpuExt_getWorkVector.argtypes = [P, P, C.c_int, P]
    lib.tfqmrgpuExt_multiply.argtypes = [P, C.c_char, I, I, C.c_uint32, P, P, P, P, P]
    lib.tfqmrgpuExt_multiplyPrepare.argtypes = [P, C.c_char, I, I, C.c_uint32, P, P, I, C.POINTER(C.c_void_p)]
    lib.tfqmrgpuExt_multiplyOrdered.argtypes = [P, C.c_char, I, I, C.c_uint32, P, P, P, P, P, P]
    lib.tfqmrgpuExt_multiplyRelease.argtypes = [P]
    lib.tfqmrgpuExt_applyOperator.argtypes = [P, P, I]
    lib.tfqmrgpuExt_shardColumns.argtypes = [I, P, I, P, P, I, P, I, I, I, C.POINTER(Shard)]
    lib.tfqmrgpuExt_freeShard.argtypes = [C.POINTER(Shard)]
    lib.tfqmrgpuExt_freeShard.restype = None
    lib.tfqmrgpuExt_commUniqueId.argtypes = [P]
    lib.tfqmrgpuExt_commInit.argtypes = [P, I, I, P]
    lib.tfqmrgpuExt_commDestroy.argtypes = [P]
    lib.tfqmrgpuExt_setReduceCallback.argtypes = [P, REDUCE_CB, P]
    lib.tfqmrgpuExt_setOperator.argtypes = [P, OPERATOR_CB, P]
    lib.tfqmrgpuExt_getRefinementHistory.argtypes = [P, P, P, C.c_int32]
    lib.tfqmrgpuExt_setThreeProductMultiply.argtypes = [P, I]
    return lib


lib = load_library()


def decode(status):
    """(code, line, char) of a status word (include/tfqmrgpu.h)."""
    key = status // CODE_CHAR
    rest = status - key * CODE_CHAR
    return rest % CODE_LINE, rest // CODE_LINE, key


def error_string(status):
    return lib.tfqmrgpuGetErrorString(status).decode()


class TfqmrError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        super().__init__("%s returned %d: %s" % (where, status, error_string(status)))


def _check(status, where, allowed=(0,)):
    if status not in allowed:
        raise TfqmrError(status, where)
    return status


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class Problem:
    """A*X==B in BSR form: index arrays plus complex block values (C row-major blocks
    A[nnzbA, LM, LM], B[nnzbB, LM, LN]); the shape the reference's readers produce (bsr.hxx:10-24)."""

    def __init__(self, rowPtrA, colIndA, A, rowPtrX, colIndX, rowPtrB, colIndB, B, X=None, tolerance=1e-9, index_offset=0):
        self.rowPtrA, self.colIndA = _i32(rowPtrA), _i32(colIndA)
        self.rowPtrX, self.colIndX = _i32(rowPtrX), _i32(colIndX)
        self.rowPtrB, self.colIndB = _i32(rowPtrB), _i32(colIndB)
        self.A = np.ascontiguousarray(A, dtype=np.complex128)
        self.B = np.ascontiguousarray(B, dtype=np.complex128)
        self.X = None if X is None else np.ascontiguousarray(X, dtype=np.complex128)
        self.tolerance = float(tolerance)
        self.index_offset = int(index_offset)
        self.mb = len(self.rowPtrA) - 1
        self.LM = self.A.shape[1]
        self.LN = self.B.shape[2]
        self.nnzbA, self.nnzbX, self.nnzbB = len(self.colIndA), len(self.colIndX), len(self.colIndB)


class Solver:
    """Staged use of the C-ABI: createHandle, setStream, createPlan, bufferSize, (workspace), setBuffer,
    setMatrix, solve, getInfo, getMatrix -- the call sequence of the reference's benchmark driver
    (bench_tfqmrgpu.cu:64-217)."""

    def __init__(self, stream=None):
        self.handle = C.c_void_p(None)
        _check(lib.tfqmrgpuCreateHandle(C.byref(self.handle)), "tfqmrgpuCreateHandle")
        _check(lib.tfqmrgpuSetStream(self.handle, C.c_void_p(stream or 0)), "tfqmrgpuSetStream")
        self.plan = C.c_void_p(None)
        self.buffer = C.c_void_p(None)
        self._own_buffer = False
        self._keep = []

    # -- plan ------------------------------------------------------------------------------------------
    def create_plan(self, pr, echo=0):
        self.problem = pr
        st = lib.tfqmrgpu_bsrsv_createPlan(self.handle, C.byref(self.plan), pr.mb,
                                           _ptr(pr.rowPtrA), pr.nnzbA, _ptr(pr.colIndA),
                                           _ptr(pr.rowPtrX), pr.nnzbX, _ptr(pr.colIndX),
                                           _ptr(pr.rowPtrB), pr.nnzbB, _ptr(pr.colIndB), pr.index_offset, echo)
        return _check(st, "tfqmrgpu_bsrsv_createPlan")

    def plan_view(self):
        v = PlanView()
        _check(lib.tfqmrgpuExt_planView(self.plan, C.byref(v)), "tfqmrgpuExt_planView")
        n = int(v.nPairs)
        return dict(nRows=v.nRows, nCols=v.nCols, nnzbA=v.nnzbA, nnzbX=v.nnzbX, nnzbB=v.nnzbB, nPairs=n,
                    pairs=np.ctypeslib.as_array(v.pairs, (2 * n,)).copy() if n else np.zeros(0, np.uint32),
                    starts=np.ctypeslib.as_array(v.starts, (v.nnzbX + 1,)).copy(),
                    subset=np.ctypeslib.as_array(v.subset, (v.nnzbB,)).copy() if v.nnzbB else np.zeros(0, np.uint32),
                    colindx=np.ctypeslib.as_array(v.colindx, (v.nnzbX,)).copy(),
                    original_bsrColIndX=np.ctypeslib.as_array(v.original_bsrColIndX, (v.nCols,)).copy())

    def buffer_size(self, LM, LN, precision):
        n = C.c_size_t(0)
        _check(lib.tfqmrgpu_bsrsv_bufferSize(self.handle, self.plan, LM, LM, LN, LN, precision.encode(), C.byref(n)),
               "tfqmrgpu_bsrsv_bufferSize")
        self.LM, self.LN, self.precision = LM, LN, precision
        # precision of the host arrays handed to set_matrix / returned by get_matrix: the plan's, for a mixed-precision plan ('m',
        # which takes both) double unless the caller sets data_precision = "c"
        self.data_precision = "z" if precision in "zm" else "c"
        return n.value

    def set_shadow_mode(self, mode):
        _check(lib.tfqmrgpuExt_setShadowMode(self.plan, mode), "tfqmrgpuExt_setShadowMode")

    def set_buffer(self, device_ptr=None, nbytes=None):
        if device_ptr is None:
            _check(lib.tfqmrgpuCreateWorkspace(C.byref(self.buffer), nbytes, b"d"), "tfqmrgpuCreateWorkspace")
            self._own_buffer = True
        else:
            self.buffer = C.c_void_p(device_ptr)
        _check(lib.tfqmrgpu_bsrsv_setBuffer(self.handle, self.plan, self.buffer), "tfqmrgpu_bsrsv_setBuffer")

    def get_shadow_vector(self):
        """the shadow vector v3 in use: float [nnzbX, 2, LM, LN], caller's block order"""
        v3 = np.zeros((self.problem.nnzbX, 2, self.LM, self.LN), dtype=np.float32)
        _check(lib.tfqmrgpuExt_getShadowVector(self.handle, self.plan, _ptr(v3)), "tfqmrgpuExt_getShadowVector")
        return v3

    def get_work_vector(self, which):
        """work vector `which` (1: X, 4 ... 9: v4 ... v9, tfqmrgpu_core.hxx:52-59) as the last solve left it: complex [nnzbX, LM, LN]"""
        v = np.zeros((self.problem.nnzbX, 2, self.LM, self.LN), dtype=self._real_dtype())
        _check(lib.tfqmrgpuExt_getWorkVector(self.handle, self.plan, which, _ptr(v)), "tfqmrgpuExt_getWorkVector")
        return v[:, 0].astype(np.float64) + 1j * v[:, 1]

    # -- values ----------------------------------------------------------------------------------------
    def _real_dtype(self):
        return np.float64 if self.data_precision == "z" else np.float32

    def set_matrix(self, var, blocks, trans="n", layout=LAYOUT_RIRIRIRI):
        """blocks: complex array [nnzb, rows, cols] (interleaved layout) or a raw real array for other layouts"""
        a = np.asarray(blocks)
        if np.iscomplexobj(a):
            a = np.ascontiguousarray(a.astype(np.complex128 if self.data_precision == "z" else np.complex64))
        else:
            a = np.ascontiguousarray(a, dtype=self._real_dtype())
        self._keep.append(a)
        st = lib.tfqmrgpu_bsrsv_setMatrix(self.handle, self.plan, var.encode(), _ptr(a), self.data_precision.encode(),
                                          self.LN if var in "XBxb" else self.LM, self.LM, trans.encode(), layout)
        return _check(st, "tfqmrgpu_bsrsv_setMatrix('%s')" % var)

    def set_matrix_device(self, var, device_ptr, trans="n", layout=LAYOUT_RIRIRIRI):
        """the same call with an array that lives in DEVICE memory (data_precision says float or double): the library converts
        straight from it, asynchronously on the handle's stream"""
        st = lib.tfqmrgpu_bsrsv_setMatrix(self.handle, self.plan, var.encode(), C.c_void_p(device_ptr), self.data_precision.encode(),
                                          self.LN if var in "XBxb" else self.LM, self.LM, trans.encode(), layout)
        return _check(st, "tfqmrgpu_bsrsv_setMatrix('%s', device array)" % var)

    def get_matrix_device(self, device_ptr, trans="n", layout=LAYOUT_RIRIRIRI):
        st = lib.tfqmrgpu_bsrsv_getMatrix(self.handle, self.plan, b"X", C.c_void_p(device_ptr), self.data_precision.encode(),
                                          self.LN, self.LM, trans.encode(), layout)
        return _check(st, "tfqmrgpu_bsrsv_getMatrix(device array)")

    def get_matrix(self, nnzb=None, trans="n", layout=LAYOUT_RIRIRIRI, raw=False):
        nnzb = self.problem.nnzbX if nnzb is None else nnzb
        out = np.zeros((nnzb, self.LM, self.LN, 2), dtype=self._real_dtype())
        st = lib.tfqmrgpu_bsrsv_getMatrix(self.handle, self.plan, b"X", _ptr(out), self.data_precision.encode(),
                                          self.LN, self.LM, trans.encode(), layout)
        _check(st, "tfqmrgpu_bsrsv_getMatrix")
        if raw or layout != LAYOUT_RIRIRIRI:
            return out.reshape(nnzb, -1)
        c = out[..., 0] + 1j * out[..., 1]
        return c if trans in "n*" else c.reshape(nnzb, self.LN, self.LM)

    def apply_operator(self, repetitions=1):
        """X := A*X on the plan's data (tfqmrgpuExt_applyOperator)"""
        return _check(lib.tfqmrgpuExt_applyOperator(self.handle, self.plan, repetitions), "tfqmrgpuExt_applyOperator")

    # -- solve -----------------------------------------------------------------------------------------
    def solve(self, threshold, max_iterations):
        """returns the raw status: 0 converged, 9 max iterations, 6 breakdown (tfqmrgpu_core.hxx:170,258,297)"""
        st = lib.tfqmrgpu_bsrsv_solve(self.handle, self.plan, threshold, max_iterations)
        return _check(st, "tfqmrgpu_bsrsv_solve", allowed=(0, 6, 9))

    def set_operator(self, multiply):
        """user-defined operator (tfqmrgpu_ext.h section 5): multiply(Y_ptr, X_ptr, colindx_ptr, nnzbX, nCols, lm, ln,
        precision, stream) enqueues Y = A*X on device data in the caller's block order and returns the flop count;
        None restores the built-in block-sparse operator"""
        if multiply is None:
            self._operator = OPERATOR_CB(0)
        else:
            def thunk(ctx, y, x, colindx, nnzbX, nCols, lm, ln, precision, stream, flops):
                try:
                    flops[0] = float(multiply(y, x, colindx, nnzbX, nCols, lm, ln, precision.decode(), stream) or 0.)
                    return 0
                except Exception:                   # an exception must not unwind through the C frames
                    import traceback; traceback.print_exc()
                    return 14
            self._operator = OPERATOR_CB(thunk)     # keep the thunk alive as long as the plan may call it
        _check(lib.tfqmrgpuExt_setOperator(self.plan, self._operator, None), "tfqmrgpuExt_setOperator")

    def get_info(self):
        r, f, fa, it = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int32(0)
        _check(lib.tfqmrgpu_bsrsv_getInfo(self.handle, self.plan, C.byref(r), C.byref(it), C.byref(f), C.byref(fa)),
               "tfqmrgpu_bsrsv_getInfo")
        return dict(residual=r.value, iterations=it.value, flops=f.value, flops_all=fa.value)

    def refinement_history(self, with_iterations=False):
        """mixed precision: relative residual (double arithmetic) in front of every float solve and at the end
        [, the float iterations of every solve]"""
        n = lib.tfqmrgpuExt_getRefinementHistory(self.plan, None, None, 0)
        h, it = np.zeros(max(n, 0), dtype=np.float64), np.zeros(max(n, 0), dtype=np.int32)
        if n > 0:
            lib.tfqmrgpuExt_getRefinementHistory(self.plan, _ptr(h), _ptr(it), n)
        return (h, it) if with_iterations else h

    def set_three_product_multiply(self, on=True):
        _check(lib.tfqmrgpuExt_setThreeProductMultiply(self.plan, int(on)), "tfqmrgpuExt_setThreeProductMultiply")

    def bound_history(self):
        n = lib.tfqmrgpuExt_getBoundHistory(self.plan, None, 0)
        h = np.zeros(max(n, 0), dtype=np.float64)
        if n > 0:
            lib.tfqmrgpuExt_getBoundHistory(self.plan, _ptr(h), n)
        return h

    PROFILE_CLASSES = ["dec35", "xpay_v6", "spmm_v4_dot", "dec34", "v5_nrm", "decT_c67", "x_v6_v7",
                       "spmm_v5_nrm_dot", "decT_final", "decide", "probe"]

    def set_profiling(self, on=True):
        _check(lib.tfqmrgpuExt_setProfiling(self.plan, int(on)), "tfqmrgpuExt_setProfiling")

    def profile(self, gated=False, first=False):
        """{kernel class: (launches, total ms)} of the last solve; gated=True: the launches that returned at once;
        first=True: of the launches that did work, those of the first iteration (which skip the operands that are zero there)"""
        n = len(self.PROFILE_CLASSES)
        cnt, ms = np.zeros(n, np.int64), np.zeros(n, np.float64)
        fn = lib.tfqmrgpuExt_getProfileGated if gated else lib.tfqmrgpuExt_getProfileFirst if first else lib.tfqmrgpuExt_getProfile
        _check(fn(self.plan, _ptr(cnt), _ptr(ms)), "tfqmrgpuExt_getProfile")
        return {k: (int(cnt[i]), float(ms[i])) for i, k in enumerate(self.PROFILE_CLASSES)}

    def multiply_kernel(self):
        """the kernel family of this plan's fused multiplies, e.g. 'k_spmm_ilv16' (needs the buffer)"""
        buf = C.create_string_buffer(64)
        _check(lib.tfqmrgpuExt_getMultiplyKernel(self.plan, buf, 64), "tfqmrgpuExt_getMultiplyKernel")
        return buf.value.decode()

    def close(self):
        if self.plan:
            lib.tfqmrgpu_bsrsv_destroyPlan(self.handle, self.plan)
            self.plan = C.c_void_p(None)
        if self._own_buffer and self.buffer:
            lib.tfqmrgpuDestroyWorkspace(self.buffer)
            self.buffer = C.c_void_p(None)
        if self.handle:
            lib.tfqmrgpuDestroyHandle(self.handle)
            self.handle = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def hash_shadow_vector(pr):
    """numpy restatement of the library's default shadow vector (tfq_device.hpp: shadow_key / shadow_quad / shadow_pick), float
    [nnzbX, 2, LM, LN] in the caller's block order -- what tests feed to the CPU oracle"""
    from .problems import _splitmix64
    rows = np.repeat(np.arange(pr.mb, dtype=np.uint64), np.diff(pr.rowPtrX))
    cols = (pr.colIndX.astype(np.int64) - pr.index_offset).astype(np.uint64)
    with np.errstate(over="ignore"):
        key = _splitmix64((cols << np.uint64(32)) | rows) ^ np.uint64(1234)
        q = np.arange((pr.LM // 2) * pr.LN, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)   # one hash per (pair of rows m, column c)
        h = _splitmix64(key[:, None] + q[None, :]).reshape(pr.nnzbX, pr.LM // 2, pr.LN)
    v = np.empty((pr.nnzbX, 2, pr.LM // 2, 2, pr.LN), dtype=np.float32)
    for odd in range(2):
        for plane in range(2):       # 16 bits each: (row 2m | 2m + 1) x (Re | Im)
            bits = (h >> np.uint64(16 * (2 * odd + plane))) & np.uint64(0xFFFF)
            v[:, plane, :, odd, :] = (bits + np.uint64(1)).astype(np.float32) * np.float32(1.0 / 65536.0)
    return v.reshape(pr.nnzbX, 2, pr.LM, pr.LN)


def solve_problem(pr, precision="z", threshold=None, max_iterations=2000, transA="n", shadow_mode=SHADOW_HASH, stream=None,
                  data_precision=None, three_products=False):
    """createPlan .. getMatrix in one go; returns (status, X[nnzbX, LM, LN] complex, info dict)."""
    with Solver(stream) as s:
        s.create_plan(pr)
        nbytes = s.buffer_size(pr.LM, pr.LN, precision)
        if data_precision:
            s.data_precision = data_precision
        if three_products:
            s.set_three_product_multiply(True)
        s.set_shadow_mode(shadow_mode)
        s.set_buffer(nbytes=nbytes)
        s.set_matrix("A", pr.A, transA)
        s.set_matrix("B", pr.B, "n")
        status = s.solve(pr.tolerance if threshold is None else threshold, max_iterations)
        info = s.get_info()
        info["bound_history"] = s.bound_history()
        info["refinement_history"] = s.refinement_history()
        info["buffer_bytes"] = nbytes
        X = s.get_matrix()
    return status, X, info


def shard_columns(pr, nranks, rank):
    """Sub-problem of `pr` owned by `rank`: contiguous range of compressed block columns of X/B
    (tfqmrgpuExt_shardColumns).  Returns (Problem, xBlocks, bBlocks)."""
    sh = Shard()
    st = lib.tfqmrgpuExt_shardColumns(pr.mb, _ptr(pr.rowPtrX), pr.nnzbX, _ptr(pr.colIndX), _ptr(pr.rowPtrB), pr.nnzbB,
                                      _ptr(pr.colIndB), pr.index_offset, nranks, rank, C.byref(sh))
    _check(st, "tfqmrgpuExt_shardColumns")
    try:
        def arr(p, n):
            return np.ctypeslib.as_array(p, (n,)).copy() if n else np.zeros(0, np.int32)
        xb, bb = arr(sh.xBlocks, sh.nnzbX), arr(sh.bBlocks, sh.nnzbB)
        off = pr.index_offset
        sub = Problem(pr.rowPtrA - off, pr.colIndA - off, pr.A,
                      arr(sh.rowPtrX, sh.mb + 1), arr(sh.colIndX, sh.nnzbX),
                      arr(sh.rowPtrB, sh.mb + 1), arr(sh.colIndB, sh.nnzbB), pr.B[bb],
                      None if pr.X is None else pr.X[xb], pr.tolerance, 0)
        sub.first_col, sub.n_cols = sh.firstCol, sh.nCols
    finally:
        lib.tfqmrgpuExt_freeShard(C.byref(sh))
    return sub, xb, bb
