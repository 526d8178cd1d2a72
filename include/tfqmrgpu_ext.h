/*
 * tfqmrgpu_ext.h -- additive extensions of the MI355X build of libtfQMRgpu.so.
 *
 * Nothing in here exists in the reference (real-space/tfQMRgpu); existing callers never
 * need it.  It adds (1) read-only views for parity tests, (2) control over the shadow
 * vector v3 so that iteration counts can be compared with the reference CPU path,
 * (3) a stand-alone BSR multiply on device-resident data (what the reference times in
 * `bench_tfqmrgpu multi`, bench_tfqmrgpu.cu:289-440) and (4) the multi-GPU mode: right
 * hand side block columns sharded over ranks, one tiny RCCL all-reduce per stopping test.
 */
#ifndef TFQMRGPU_EXT_H
#define TFQMRGPU_EXT_H

#include "tfqmrgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- (1) plan introspection ------------------------------------------------------------ */
/* The arrays are exactly the analysis results of the reference createPlan
 * (tfqmrgpu.cu:183-339, members of bsrsv_plan_t, tfqmrgpu_plan.hxx:21-49): same order,
 * same integer types, 0-based.  Pointers stay valid until destroyPlan. */
typedef struct {
    uint32_t nRows;                        /* mb                                            */
    uint32_t nCols;                        /* non-empty block columns of X                  */
    uint32_t nnzbA, nnzbX, nnzbB;
    uint64_t nPairs;                       /* block products in Y = A*X                     */
    uint32_t const *pairs;                 /* [2*nPairs]  (inzA, inzX)                      */
    uint32_t const *starts;                /* [nnzbX + 1] first pair of each Y block        */
    uint32_t const *subset;                /* [nnzbB]     X block that holds each B block   */
    uint16_t const *colindx;               /* [nnzbX]     compressed block column           */
    int32_t  const *original_bsrColIndX;   /* [nCols]     user column index per compressed  */
    int32_t  LM, LN;                       /* block shape, 0 before bufferSize              */
    char     precision;                    /* 'c' / 'z', 0 before bufferSize                */
} tfqmrgpuPlanView_t;

tfqmrgpuStatus_t tfqmrgpuExt_planView(tfqmrgpuBsrsvPlan_t plan, tfqmrgpuPlanView_t *view);

/* per-iteration trace of the last solve: bound2[it] = max_rhs(tau*invBn2)*(2*it+1) as the
 * stopping test saw it (reference tfqmrgpu_core.hxx:239-252).  Returns how many entries
 * exist; copies at most `capacity`. */
int32_t tfqmrgpuExt_getBoundHistory(tfqmrgpuBsrsvPlan_t plan, double *bound2, int32_t capacity);

/* per-kernel timing of the last solve, measured with HIP events on the solver's own stream.
 * Switch on before solve; afterwards read, for each kernel class k < TFQMRGPU_PROFILE_CLASSES,
 * the number of launches that did work and their summed duration in milliseconds. */
enum {
    TFQMRGPU_PROF_DEC35 = 0, TFQMRGPU_PROF_XPAY_V6, TFQMRGPU_PROF_SPMM_V4_DOT, TFQMRGPU_PROF_DEC34,
    TFQMRGPU_PROF_V5_NRM, TFQMRGPU_PROF_DECT_C67, TFQMRGPU_PROF_X_V6_V7, TFQMRGPU_PROF_SPMM_V5_NRM_DOT,
    TFQMRGPU_PROF_DECT_FINAL, TFQMRGPU_PROF_DECIDE, TFQMRGPU_PROF_PROBE,
    TFQMRGPU_PROFILE_CLASSES
};
/* on = 1: events around every kernel class (176 events per 16 iterations: costs a 37 ms solve 0.6 ms);
 * on = 2: only around the two fused multiplies SPMM_V4_DOT and SPMM_V5_NRM_DOT (the other classes report 0 launches) */
tfqmrgpuStatus_t tfqmrgpuExt_setProfiling(tfqmrgpuBsrsvPlan_t plan, int on);
tfqmrgpuStatus_t tfqmrgpuExt_getProfile(tfqmrgpuBsrsvPlan_t plan, int64_t *launches, double *milliseconds);
/* the launches that were enqueued ahead of the stopping decision and returned without doing work
 * (a profiler counts them as calls of the same kernels) */
tfqmrgpuStatus_t tfqmrgpuExt_getProfileGated(tfqmrgpuBsrsvPlan_t plan, int64_t *launches, double *milliseconds);
/* of the launches that getProfile counts: those of the FIRST iteration of the solve.  v4, v6, v7, v8 and x are zero there and v5 is
 * B scattered onto zeros by definition (tfqmrgpu_core.hxx:125,147-153); they are neither written at the start of a solve nor
 * read: XPAY_V6, SPMM_V4_DOT, V5_NRM and X_V6_V7 move 2, 2, 1 and 2 vectors less in that launch -- price a kernel against its
 * roof on the other launches. */
tfqmrgpuStatus_t tfqmrgpuExt_getProfileFirst(tfqmrgpuBsrsvPlan_t plan, int64_t *launches, double *milliseconds);
/* the kernel family that the fused multiplies of this plan run (needs the buffer: the element order and the column batches are fixed
 * by bufferSize / setBuffer), e.g. "k_spmm_ilv16" -- so that kept profiler figures (profiles/pmc_traffic.json) can be told apart from
 * figures of a kernel that no longer runs.  Writes at most `capacity` bytes including the terminating 0. */
tfqmrgpuStatus_t tfqmrgpuExt_getMultiplyKernel(tfqmrgpuBsrsvPlan_t plan, char *name, int32_t capacity);

/* ---- (2) shadow vector v3 -------------------------------------------------------------- */
enum {
    TFQMRGPU_SHADOW_HASH       = 0, /* default: counter-based hash of (block row, block column,
                                       element), uniform (0,1]; independent of the GPU count  */
    TFQMRGPU_SHADOW_GLIBC_RAND = 1  /* the sequence the reference CPU path uses: glibc rand()
                                       from seed 1, /RAND_MAX, flat over [nnzbX][2][LM][LN]
                                       (tfqmrgpu_linalg.hxx:799-802)                          */
};
/* call after createPlan and before setBuffer */
tfqmrgpuStatus_t tfqmrgpuExt_setShadowMode(tfqmrgpuBsrsvPlan_t plan, int mode);
/* user-supplied v3, host array float[nnzbX][2][LM][LN] in the caller's BSR order; call after setBuffer */
tfqmrgpuStatus_t tfqmrgpuExt_setShadowVector(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, float const *v3);
/* the vector in use, same shape and order (for parity tests: feed it to a CPU implementation); call after setBuffer,
 * not during a solve.  In hash mode the four reals (Re, Im) x (row 2m, row 2m+1) of column j of the block in block row r,
 * ORIGINAL block column c (0-based) come from one 64-bit hash, 16 bits each:
 *   h = splitmix64(key + (m * ln + j) * 0xd1342543de82ef95),  key = splitmix64(c << 32 | r) ^ 1234,
 *   value = float(((h >> 16 * (2 * (row & 1) + (Im ? 1 : 0))) & 0xffff) + 1) / 2^16   in (0, 1]
 * (tfq_device.hpp), whatever the block order and however the columns are sharded over GPUs. */
tfqmrgpuStatus_t tfqmrgpuExt_getShadowVector(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, float *v3);

/* Debug getter for kernel-level parity tests: one of the solver's X-shaped work vectors as the last solve left it --
 * which = 1: the solution X, 4 ... 9: v4 ... v9 in the reference's numbering (tfqmrgpu_core.hxx:52-59; v4 = A-image
 * recurrence, v5 = residual-like vector, v6/v7 = search directions, v8 = A v6 of the second half step, v9 = A v6 of the
 * first) -- to host memory, native layout [nnzbX][2][lm][ln], caller's block order, plan precision.  Non-destructive;
 * call it before the next set/getMatrix, which stage the caller's blocks through v4 ... v9.
 * After a solve that stopped at iteration k the vectors are those of the reference at the end of iteration k, except
 * that the residual probe of this library does not overwrite v9 (the reference's does, tfqmrgpu_core.hxx:265). */
tfqmrgpuStatus_t tfqmrgpuExt_getWorkVector(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, int which, void *values);

/* ---- (3) stand-alone block-sparse multiply --------------------------------------------- */
/* Y[iY] = sum over pairs p in [starts[iY], starts[iY+1]) of A[pairs[2p]] * X[pairs[2p+1]]
 * All pointers are DEVICE pointers.  Blocks are in the native layout RRRRIIII:
 * A[nnzbA][2][lm(k)][lm(i)] (transposed), X|Y[nnzb][2][lm][ln].
 * Same contract as the reference kernel gemmNxNf (tfqmrgpu_blockmult.hxx:9-93).
 * Four real products per complex one, as the reference (the three-product form is an option of a PLAN, section 6). */
tfqmrgpuStatus_t tfqmrgpuExt_multiply(tfqmrgpuHandle_t handle,
    char precision, int lm, int ln,
    uint32_t nnzbY, uint32_t const *starts_d, uint32_t const *pairs_d,
    void const *A_d, void const *X_d, void *Y_d);

/* A PREPARED launch order for that product (r04).  The listing stays the caller's; which work group computes which Y block is the library's to
 * choose, and for a listing that is multiplied many times it is chosen once, outside the caller's timed loop -- as the reference's own
 * benchmark prepares its launch (bench_tfqmrgpu.cu:442-556 in front of the timed loop :289-440).  multiplyPrepare reads the two lists
 * back from the device, finds block columns (Y blocks that share X blocks) and row bands (by the A indices) and leaves an XCD-aware order
 * in device memory: mode 1 = neighbouring work groups share X and A blocks, the 8 XCDs split the block COLUMNS (every L2 sees all of A and an
 * eighth of X); mode 3 = the XCDs split the block ROWS (an eighth of A, all of X); mode 4 = 1 or 3, whichever keeps the larger operand split
 * (the recommended one); mode 2 = mode 1 with the work groups of most block products first; mode 0 or a shape whose kernel takes no order: *order = NULL,
 * which multiplyOrdered treats as the caller's order.  Results are those of tfqmrgpuExt_multiply bit for bit (the same kernel computes
 * every Y block from the same pair list).  An order belongs to ONE listing (nnzbY, starts, pairs): release it with multiplyRelease. */
tfqmrgpuStatus_t tfqmrgpuExt_multiplyPrepare(tfqmrgpuHandle_t handle, char precision, int lm, int ln,
    uint32_t nnzbY, uint32_t const *starts_d, uint32_t const *pairs_d, int mode, void **order);
tfqmrgpuStatus_t tfqmrgpuExt_multiplyOrdered(tfqmrgpuHandle_t handle,
    char precision, int lm, int ln,
    uint32_t nnzbY, uint32_t const *starts_d, uint32_t const *pairs_d,
    void const *A_d, void const *X_d, void *Y_d, void const *order);
tfqmrgpuStatus_t tfqmrgpuExt_multiplyRelease(void *order);

/* The same product on the data of a plan: X := A * X for the plan's operator A (as given to setMatrix('A')) and the
 * plan's X (setMatrix('X') before, getMatrix('X') afterwards), truncated to the pattern of X like every product of the
 * solver.  Uses the multiply kernel and the block / element order of the solver itself -- for 16 x 16
 * complex<double> plans the row-pair-interleaved one -- so it is also what bench.py times as "the BSR multiply".
 * `repetitions` > 1 computes the product that many times from the same X (timing); `repetitions` < 0: that many launches and NO copy
 * of the product back into X (a timed region then holds the multiply kernel alone); asynchronous on the handle's stream. */
tfqmrgpuStatus_t tfqmrgpuExt_applyOperator(tfqmrgpuHandle_t handle, tfqmrgpuBsrsvPlan_t plan, int repetitions);

/* ---- (4) multi-GPU: one process per GPU, block columns of X/B sharded ------------------- */
/* Splits the compressed block columns of X into `nranks` contiguous ranges with balanced
 * block counts and extracts the sub-patterns of X and B that belong to `rank`.
 * Outputs are malloc'ed by the library; release with tfqmrgpuExt_freeShard.
 * xBlocks/bBlocks list, for every block of the shard, its index in the unsharded operator
 * (so that values can be scattered/gathered).  Index arrays are 0-based regardless of
 * indexOffset of the input. */
typedef struct {
    int32_t  mb;
    int32_t  nnzbX, nnzbB;
    int32_t *rowPtrX, *colIndX;  /* [mb+1], [nnzbX] */
    int32_t *rowPtrB, *colIndB;  /* [mb+1], [nnzbB] */
    int32_t *xBlocks, *bBlocks;  /* [nnzbX], [nnzbB] indices into the global value arrays   */
    int32_t  firstCol, nCols;    /* range of compressed block columns owned by this rank     */
} tfqmrgpuShard_t;

tfqmrgpuStatus_t tfqmrgpuExt_shardColumns(int mb,
    int32_t const *rowPtrX, int nnzbX, int32_t const *colIndX,
    int32_t const *rowPtrB, int nnzbB, int32_t const *colIndB,
    int indexOffset, int nranks, int rank, tfqmrgpuShard_t *shard);
void tfqmrgpuExt_freeShard(tfqmrgpuShard_t *shard);

/* RCCL communicator for the stopping test.  rank 0 creates the id, the caller broadcasts the
 * 128 bytes by any means (torch.distributed, MPI, a file), every rank calls commInit. */
tfqmrgpuStatus_t tfqmrgpuExt_commUniqueId(char id[128]);
tfqmrgpuStatus_t tfqmrgpuExt_commInit(tfqmrgpuHandle_t handle, int nranks, int rank, char const id[128]);
tfqmrgpuStatus_t tfqmrgpuExt_commDestroy(tfqmrgpuHandle_t handle);
/* Instead of RCCL: a host callback that max-reduces `n` doubles in place over all ranks
 * (used by the CPU/gloo tests of the sharding logic and by MPI-based callers). */
typedef void (*tfqmrgpuReduceMax_t)(void *ctx, double *values, int n);
tfqmrgpuStatus_t tfqmrgpuExt_setReduceCallback(tfqmrgpuHandle_t handle, tfqmrgpuReduceMax_t fn, void *ctx);

/* ---- (6) precision options -------------------------------------------------------------- */
/* Mixed precision: tfqmrgpu_bsrsv_bufferSize(..., 'm', ...) -- dormant in the reference (tfqmrgpu.cu:42, "load float, multiply-
 * accumulate double, store float"; tfqmrgpu.h:72 "start with float and converge double"), built here as iterative refinement:
 * x, B and A are kept in double, every cycle computes r = b - A x in double, solves A d = r with the complex<float> tfQMR and adds
 * d to x in double.  setMatrix / getMatrix of such a plan accept 'c' AND 'z' data (converted on the way); solve's threshold is
 * max_rhs |b - A x| / |b| in double arithmetic, maxIterations bounds the sum of the float iterations; getInfo reports that sum.
 * The buffer is 11 float-sized vectors against 15 for 'z'.  Where float iterations cannot reduce the residual (systems on which
 * the 'c' solver stagnates above ~0.1) solve returns TFQMRGPU_STATUS_MAX_ITERATIONS with the best x.
 * getRefinementHistory: residual[i] = the relative residual (double arithmetic) in front of float solve i, the last entry the final
 * one; iterations[i] = the float iterations of solve i (0 in the last entry); either array may be NULL; returns the count. */
int32_t tfqmrgpuExt_getRefinementHistory(tfqmrgpuBsrsvPlan_t plan, double *residual, int32_t *iterations, int32_t capacity);
/* Three real products per complex one (Gauss) in the complex<double> multiplies of the block shapes above 16 x 16: a quarter fewer
 * matrix instructions (64 x 64: iteration -7 %), but Im = P3 - P1 - P2 is accurate relative to |A||X| only -- an imaginary part
 * 10^-k times smaller than the real part loses k digits against the reference's four products.  OFF unless switched on here
 * (it was the default until round 2); call before solve. */
tfqmrgpuStatus_t tfqmrgpuExt_setThreeProductMultiply(tfqmrgpuBsrsvPlan_t plan, int on);

/* ---- (5) user-defined linear operator --------------------------------------------------- */
/* The reference lets C++ users replace the block-sparse operator by their own `action_t` class whose
 * `multiply(y, x, colindx, nnzbX, nCols, l2nX, streamId)` returns the flop count (README.md:110-117,
 * tfqmrgpu_blocksparse.hxx:71-199, called at tfqmrgpu_core.hxx:134).  The C-ABI counterpart: a callback
 * that ENQUEUES Y = A*X on `stream` (no synchronisation) for device vectors in the caller's own BSR block
 * order of X and the native layout Y|X[nnzbX][2][lm][ln]; colindx_d[nnzbX] is the compressed block column
 * of every block (what the reference hands over).  *flops receives the operation count of the call.
 * A non-zero return value aborts the solve and is returned by tfqmrgpu_bsrsv_solve.
 * With an operator installed the solver runs its un-fused schedule (gather into the caller's order,
 * callback, vector-update kernel) and synchronises with the host once per iteration, like the reference;
 * createPlan still needs a pattern for A (any valid one, e.g. block-diagonal), its values are not used.
 * Two X-shaped scratch vectors are allocated by the library at the first solve and released by
 * destroyPlan.  multiply == NULL restores the built-in block-sparse operator. */
typedef tfqmrgpuStatus_t (*tfqmrgpuOperator_t)(void *ctx, void *Y_d, void const *X_d,
    uint16_t const *colindx_d, uint32_t nnzbX, uint32_t nCols, int lm, int ln, char precision,
    tfqmrgpuStream_t stream, double *flops);
tfqmrgpuStatus_t tfqmrgpuExt_setOperator(tfqmrgpuBsrsvPlan_t plan, tfqmrgpuOperator_t multiply, void *ctx);

#ifdef __cplusplus
}
#endif
#endif /* TFQMRGPU_EXT_H */
