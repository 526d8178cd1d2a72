/*
 * tfqmrgpu.h -- C-ABI of the MI355X-native block-sparse tfQMR solver.
 *
 * Drop-in boundary: every prototype below keeps the name, argument order and
 * argument types of the reference interface real-space/tfQMRgpu
 * `tfQMRgpu/include/tfqmrgpu.h:16-156`, so that existing C / Fortran / Julia /
 * Python(ctypes) callers relink against this libtfQMRgpu.so unchanged.
 * Each declaration cites the reference line it replaces.
 *
 * Differences a maintainer should know (all source compatible):
 *  - the header is self-contained (pulls <stdint.h>/<stddef.h>, has include
 *    guards and extern "C"); the reference header relies on its includer.
 *  - the stream argument is a HIP stream.  It travels as one pointer-sized
 *    value, exactly like the reference's cudaStream_t.  Legacy callers that
 *    already typedef their own `cudaStream_t` compile with
 *    -DTFQMRGPU_STREAM_TYPE=cudaStream_t; HIP callers may use
 *    -DTFQMRGPU_STREAM_TYPE=hipStream_t.  0/NULL selects the default stream.
 *  - status / layout constants are enumerators instead of `const` globals
 *    (a `const` global in a C header is a definition per translation unit).
 *  - the work buffer handed to tfqmrgpu_bsrsv_setBuffer is HIP device memory
 *    (hipMalloc or tfqmrgpuCreateWorkspace).
 */
#ifndef TFQMRGPU_H
#define TFQMRGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- opaque types (reference tfqmrgpu.h:5-9) ------------------------------------------ */
typedef int32_t tfqmrgpuStatus_t;      /* 0 == success, else code + 1000*line + 1e7*char   */
typedef void*   tfqmrgpuHandle_t;      /* library handle, carries the stream               */
typedef int*    tfqmrgpuBsrsvPlan_t;   /* analysis result + solver state of one A*X==B     */
typedef int     tfqmrgpuDataLayout_t;  /* how Re/Im parts of a block are interleaved       */

#ifdef TFQMRGPU_STREAM_TYPE
typedef TFQMRGPU_STREAM_TYPE tfqmrgpuStream_t;
#else
typedef void* tfqmrgpuStream_t;        /* a hipStream_t */
#endif

/* ---- error reporting (reference tfqmrgpu.h:16-17, tfqmrgpu_error_tool.cxx:33-76) ------- */
tfqmrgpuStatus_t tfqmrgpuPrintError(tfqmrgpuStatus_t const status);
char const*      tfqmrgpuGetErrorString(tfqmrgpuStatus_t const status);

/* ---- handle and stream (reference tfqmrgpu.h:20-28) ------------------------------------ */
/* *handle must be NULL on entry */
tfqmrgpuStatus_t tfqmrgpuCreateHandle(tfqmrgpuHandle_t *handle);
tfqmrgpuStatus_t tfqmrgpuDestroyHandle(tfqmrgpuHandle_t handle);
tfqmrgpuStatus_t tfqmrgpuSetStream(tfqmrgpuHandle_t handle, tfqmrgpuStream_t const streamId);
tfqmrgpuStatus_t tfqmrgpuGetStream(tfqmrgpuHandle_t handle, tfqmrgpuStream_t *streamId);

/* ---- device work space helpers (reference tfqmrgpu.h:30-31) ---------------------------- */
/* memType 'm'/'M': managed memory, anything else: device memory */
tfqmrgpuStatus_t tfqmrgpuCreateWorkspace(void* *pBuffer, size_t const pBufferSizeInBytes, char const memType);
tfqmrgpuStatus_t tfqmrgpuDestroyWorkspace(void* pBuffer);

/* ---- supported (ldA, ldB) block shapes (reference tfqmrgpu.h:33-38) -------------------- */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_allowedBlockSizes(
    int32_t *number,       /* out: how many (ldA,ldB) pairs exist                          */
    int32_t *blockSizes,   /* out: pairs, written while they fit                           */
    int const arrayLength);/* in:  capacity of blockSizes[] in int32 entries               */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_blockSizeMissing(int const ldA, int const ldB); /* 0: supported */

/* ---- bsrsv: solve A*X == B, all three operators block-compressed-sparse-row ------------ */
/* call order: createPlan, bufferSize, (allocate), setBuffer, setMatrix A, setMatrix B,
 *             solve, getInfo, getMatrix X, destroyPlan                                     */

/* index analysis; *plan must be NULL on entry (reference tfqmrgpu.h:47-60) */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_createPlan(tfqmrgpuHandle_t handle,
    tfqmrgpuBsrsvPlan_t *plan,
    int     const mb,            /* block rows of A, X, B == block columns of A            */
    int32_t const *bsrRowPtrA,   /* [mb+1]                                                 */
    int     const nnzbA,
    int32_t const *bsrColIndA,   /* [nnzbA]                                                */
    int32_t const *bsrRowPtrX,   /* [mb+1]                                                 */
    int     const nnzbX,
    int32_t const *bsrColIndX,   /* [nnzbX]                                                */
    int32_t const *bsrRowPtrB,   /* [mb+1]                                                 */
    int     const nnzbB,         /* pattern of B must be contained in that of X            */
    int32_t const *bsrColIndB,   /* [nnzbB]                                                */
    int     const indexOffset,   /* 0: C indices, 1: Fortran indices                       */
    int     const echo);         /* verbosity 0..9                                         */

/* reference tfqmrgpu.h:62-63 */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_destroyPlan(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan);

/* fixes block shape + precision in the plan, reports the device bytes the solver needs
 * (reference tfqmrgpu.h:66-73) */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_bufferSize(tfqmrgpuHandle_t handle,
    tfqmrgpuBsrsvPlan_t plan,
    int const ldA,               /* leading dimension of A blocks                          */
    int const blockDim,          /* A blocks are blockDim x blockDim, must equal ldA        */
    int const ldB,               /* leading dimension of X and B blocks, ldB >= ldA         */
    int const RhsBlockDim,       /* right-hand sides per block, must equal ldB              */
    char const precision,        /* 'c' complex<float>, 'z' complex<double>                 */
    size_t *pBufferSizeInBytes);

/* registers the device buffer, fills the shadow vector, uploads index lists
 * (reference tfqmrgpu.h:79-85) */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_setBuffer(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, void* const pBuffer);
tfqmrgpuStatus_t tfqmrgpu_bsrsv_getBuffer(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, void* *pBuffer);

/* upload block values of 'A', 'B' (or 'X'); reference tfqmrgpu.h:87-95.
 * `val` may also be DEVICE memory (hipMalloc / hipMallocManaged; setMatrix and getMatrix alike): the blocks are then converted
 * straight from / into the caller's array by one kernel on the handle's stream, without staging and without crossing PCIe --
 * the call returns when the kernel is enqueued, the array must stay valid until the stream has passed it. */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_setMatrix(tfqmrgpuHandle_t handle,
    tfqmrgpuBsrsvPlan_t plan,
    char const var,              /* 'A', 'B' or 'X'                                        */
    void const *val,             /* host values, float or double according to precision    */
    char const precision,        /* must match the plan: 'c' or 'z'                         */
    int const ld,                /* not used (blocks are dense)                             */
    int const d2,                /* not used                                                */
    char const trans,            /* 'n', 't', 'c'/'h' (conj. transpose), '*' (conjugate)    */
    tfqmrgpuDataLayout_t const layout);

/* download block values, only 'X'; reference tfqmrgpu.h:97-105 */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_getMatrix(tfqmrgpuHandle_t handle,
    tfqmrgpuBsrsvPlan_t plan,
    char const var,
    void       *val,
    char const precision,
    int const ld,
    int const d2,
    char const trans,
    tfqmrgpuDataLayout_t const layout);

/* run the transpose-free QMR iteration; returns 0, 9 (max iterations) or 6 (breakdown)
 * (reference tfqmrgpu.h:107-110) */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_solve(tfqmrgpuHandle_t handle,
    tfqmrgpuBsrsvPlan_t plan,
    double const threshold,      /* relative residual ||A x - b|| / ||b|| per right-hand side */
    int const maxIterations);

/* reference tfqmrgpu.h:112-117; any pointer may be NULL, all NULL -> status 3 */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_getInfo(tfqmrgpuHandle_t handle,
    tfqmrgpuBsrsvPlan_t plan,
    double  *residuum_reached,
    int32_t *iterations_needed,
    double  *flops_performed,
    double  *flops_performed_all);

/* one-call drivers, blocks interleaved Re/Im:  A[nnzbA][ldA][ldA][2], X|B[nnzb][ldA][ldB][2]
 * (reference tfqmrgpu.h:138-156) */
tfqmrgpuStatus_t tfqmrgpu_bsrsv_z(
    int mb, int ldA, int ldB,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA, double const* Amat, char transA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX, double      * Xmat, char transX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB, double const* Bmat, char transB,
    int32_t *iterations,   /* in: max iterations, out: iterations needed                    */
    float   *residual,     /* in: threshold,      out: residual reached                     */
    int indexOffset, int echo);

tfqmrgpuStatus_t tfqmrgpu_bsrsv_c(
    int mb, int ldA, int ldB,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA, float const* Amat, char transA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX, float      * Xmat, char transX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB, float const* Bmat, char transB,
    int32_t *iterations, float *residual, int indexOffset, int echo);

/* ---- constants (reference tfqmrgpu.h:160-191) ------------------------------------------ */
enum {
    TFQMRGPU_STATUS_SUCCESS           =  0,
    TFQMRGPU_STATUS_LAUNCH_FAILED     =  2,
    TFQMRGPU_STATUS_NO_INFO_PASSED    =  3,
    TFQMRGPU_STATUS_ALLOCATION_FAILED =  4,
    TFQMRGPU_STATUS_RANDOM_GEN_FAILED =  5,
    TFQMRGPU_STATUS_BREAKDOWN         =  6,
    TFQMRGPU_POINTER_INVALID          =  7,
    TFQMRGPU_STATUS_MAX_ITERATIONS    =  9,
    TFQMRGPU_B_HAS_A_ZERO_COLUMN      = 11,
    TFQMRGPU_BLOCKSIZE_MISSING        = 12,
    TFQMRGPU_B_IS_NOT_SUBSET_OF_X     = 13,
    TFQMRGPU_UNDOCUMENTED_ERROR       = 14,
    TFQMRGPU_DATALAYOUT_UNKNOWN       = 15,
    TFQMRGPU_PRECISION_MISSMATCH      = 16,
    TFQMRGPU_TANSPOSITION_UNKNOWN     = 17,
    TFQMRGPU_VARIABLENAME_UNKNOWN     = 18,
    TFQMRGPU_NO_IMPLEMENTATION        = 19,
    /* a status packs  code + TFQMRGPU_CODE_LINE*line + TFQMRGPU_CODE_CHAR*character */
    TFQMRGPU_CODE_LINE                = 1000,
    TFQMRGPU_CODE_CHAR                = 10000*1000
};

enum { /* bit pattern over a 2x2 complex block: 0 = real part, 1 = imaginary part */
    TFQMRGPU_LAYOUT_RRRRIIII = 0x0f,  /* device-native: Re plane then Im plane per block     */
    TFQMRGPU_LAYOUT_RRIIRRII = 0x33,  /* per block row: Re row then Im row                   */
    TFQMRGPU_LAYOUT_RIRIRIRI = 0x55   /* interleaved, C/Fortran complex                      */
};

#define TFQMRGPU_MEMORY_ALIGNMENT 8            /* log2 of the 256-byte buffer granule     */
#define TFQMRGPU_NUMBER_OF_INSTANCES_OF_X 7

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* TFQMRGPU_H */
