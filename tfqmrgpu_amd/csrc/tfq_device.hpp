// Structures shared between the host driver and the HIP kernels.
#pragma once
#include <cstdint>
#include <cstddef>
#include <hip/hip_runtime.h>

#include "tfq_plan.hpp"

namespace tfq {

// Device-resident control block of one solve.  The stopping logic of the reference runs on the
// host after two blocking copies per iteration (tfqmrgpu_core.hxx:235-304); here it runs in tiny
// kernels that update this record, every other kernel looks at `state`/`probe` first and
// returns when it has nothing to do.  The host only reads a copy of it, a few iterations late.
struct Ctl {
    double tol2;               // threshold^2
    double target_bound2;      // probe when bound2 <= target_bound2
    double max_bound2;         // bound2 of the latest iteration: max_rhs(tau/|b|^2) * (2 it + 1)
    double residual2_reached;  // max_rhs |A x - b|^2/|b|^2 at the latest probe
    double red[6];             // max-reduced over ranks: {max tau/|b|^2, any RHS alive, a rank failed} per iteration,
                               //                          {max res^2, any RHS unconverged, a rank failed} per probe
    double probe_bound2;       // max_bound2 at the latest probe (0: no probe yet): what the recurrence promised since then, for the stall test
    int32_t iteration;         // completed iterations
    int32_t maxIterations;
    int32_t state;             // 0 running, 1 converged, 2 all RHS broke down, 3 out of iterations, 4 stopped: a rank reported a failure
    int32_t probe;             // the true residual has to be computed now
    int32_t iterations_needed;
    int32_t nprobes;
    int32_t xpend;             // x += eta2*v7 of the last iteration has not been applied yet
    int32_t stallStop;         // inner solves of the mixed-precision mode: a probe that finds the true residual no better than 0.7 x the
                               // previous probe's -- or improved by 3 x less than the recurrence's bound since then -- ends the solve
                               // (state 3): the float iteration has reached its floor
};
static_assert(sizeof(Ctl) == 120, "Ctl is copied as a whole");

// device pointers of one plan (all inside the user's work buffer)
struct DevPlan {
    int LM, LN;
    bool dbl;
    uint32_t nCols, nnzbX, nnzbB, nnzbA, nChunks;
    int hashV3;                                // v3 holds the counter-based hash below: kernels may recompute instead of reading it
    int ilv;                                   // element order inside a block plane: 0 native [r][s], G = 2 | 4: groups of G rows interleaved (see ilv_offset)
    int aOnce;                                 // a multiply uses every A block about once (<= 1.5 times): A is streamed, not cached
    int first;                                 // set by the host for the launches of the FIRST iteration of a solve: v4, v6, v7, v8 and x are
                                               // zero and v5 is B scattered onto zeros by definition there (tfqmrgpu_core.hxx:125,147-153) --
                                               // the kernels take that as given instead of reading it, and the start of a solve writes none of
                                               // them (13 S of traffic per solve)
    void *x, *v4, *v5, *v6, *v7, *v8, *v9, *B, *A;
    float* v3;
    void *rho, *alfa, *beta, *c67, *eta;       // [nCols][2][LN] real
    void *c67a, *eta2;                         // dec34's c67 / the eta of the second half step
    double *z, *d, *tau, *var, *invBn2;        // [nCols][2|1][LN]
    int8_t* status;                            // [nCols][LN]
    Ctl* ctl;
    double *pz, *pd;                           // [nChunks][2|1][LN]
    double* colrec;                            // [nCols][2]
    double* colPart; uint32_t colSegMax;       // long columns are summed by several work groups (tfq_colops.hpp: column_total):
                                               // their shares [slot][3][LN], and the number of segments the longest column needs (the grid's y extent)
    uint32_t const *chunkFirst, *chunkCol, *colChunkPtr, *colStart, *bOfX, *order;
    uint32_t const* orderB; uint32_t nChunksB;  // launch order and size of the batched multiply: the chunks of the batches' first columns
    uint8_t const* colBatch;                   // not null: (batch size << 4) | position per block column -- columns with identical row patterns, multiplied together (k_spmm_ilv8b)
    uint32_t const *starts, *pairs, *subset, *bColPtr, *bList, *u2i, *rowI;
    int32_t const* origCol;
    void const* R;                             // not null: the right-hand side of THIS solve is the X-shaped vector R (the residual of the
                                               // mixed-precision refinement) instead of B scattered onto zeros; its |r|^2 per right-hand side
                                               // has been set up by k_refine_init_col
    int m3;                                    // three real products per complex one in the double multiplies above 16 x 16 (opt-in)
    int fold;                                  // small systems: the column operations run in the tail of the kernel that produces their
                                               // input (tfq_colops.hpp) instead of in launches of their own
    uint32_t* foldCount;                       // [nCols + 1] arrival counters of that scheme (zero between uses)
    DevPlan const* self;                       // a copy of this structure in device memory (for kernels that get SpmmArgs)
};

DevPlan resolve(Plan const& p);

// mixed precision 'm': the double-precision side of the plan (x, B, A, the product A x) next to the float plan `d` of the inner solves
struct RefineArgs {
    DevPlan d;                                 // the inner (float) plan: chunk tables, scalars, R = d.R
    double* xz; double const* Bz; double const* Yz;   // solution, right-hand sides, A x (all double, element order ilvZ)
    double* bn2z;                              // [nCols][LN] |b|^2, kept from cycle 0
    double* refine;                            // {max_rhs |r|^2 / |b|^2, a value was not finite, a rank failed}
    int ilvZ;
    int cycle;                                 // 0: x = 0, r = b (Yz is not read)
    double innerTol; int innerMaxIt;
};
void launch_refine_residual(RefineArgs const& r, hipStream_t s);   // r = b - A x -> R (float), |r|^2 records, inner solve set up, refine[]
void launch_refine_update(RefineArgs const& r, hipStream_t s);     // x (double) += x (float) of the inner solve

// ---- shadow vector -------------------------------------------------------------------------------
// The reference fills v3 with cuRAND XORWOW uniforms on the GPU and rand()/RAND_MAX on the CPU
// (tfqmrgpu_linalg.hxx:777-806); any positive random vector works.  Default here: a counter-based hash of
// (original block column, block row, element) -> uniform (0, 1], so the value of an element does not depend on the
// block order nor on how the columns are sharded over GPUs -- and the multiply kernels can recompute the values in
// registers instead of reading S/2 (`z`) or S (`c`) bytes per fused launch.
#ifdef __HIPCC__
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t shadow_key(uint32_t origCol, uint32_t row) { return splitmix64((uint64_t(origCol) << 32) | uint64_t(row)) ^ 1234u; }
// One 64-bit hash serves the four reals of a pair of rows: 16 bits each for (row 2m | 2m + 1) x (Re | Im) of column c of the block.
// (Round 1 and the first half of round 2 drew one hash per real: the 64-bit arithmetic cost the fused multiplies 1-6 %,
// profiles/r02_ab_hash.txt; the lanes of the interleaved kernels hold exactly these four values.)
__host__ __device__ inline uint64_t shadow_quad(uint64_t key, uint32_t m, uint32_t c, uint32_t LN) {
    return splitmix64(key + uint64_t(m * LN + c) * 0xd1342543de82ef95ull);
}
__host__ __device__ inline float shadow_pick(uint64_t h, int oddRow, int plane) {
    return float((uint32_t(h >> (16 * (2 * oddRow + plane))) & 0xffffu) + 1u) * (1.f / 65536.f);   // 16 random bits -> (0, 1]
}
// element (plane = Re | Im, row r, column c) of the block
__host__ __device__ inline float shadow_value(uint64_t key, int plane, uint32_t r, uint32_t c, uint32_t LN) {
    return shadow_pick(shadow_quad(key, r >> 1, c, LN), int(r & 1), plane);
}
#endif

// ---- element order inside one plane (Re or Im) of a block -------------------------------------------------------------
// native (the reference's, tfqmrgpu_linalg.hxx:332-345):   [r][s]            r = row (k for the transposed A blocks), s contiguous
// groups of G rows interleaved:                             [r/G][s][r%G]     G consecutive rows of one column are 16 contiguous bytes:
//                                                           G = 2 for complex<double> (16 x 16, 8 x 8), G = 4 for complex<float> (16 x 16)
// The second form exists for the multiply: a lane of the 16x16x4 MFMA then fetches the operands of two k-steps, and the
// two rows of a column that its accumulator registers hold, as ONE 16-byte access -- half the memory instructions for the
// operands, the epilogue vectors and the stores (measured on P2: fused multiplies 0.684 -> 0.628 ms, profiles/r02_lab.txt).
// Everything else (vector updates, reductions) is elementwise and only needs to know which column an element belongs to.
__host__ __device__ inline int ilv_offset(int G, int r, int s, int nC) { return ((r / G) * nC + s) * G + (r % G); }
__host__ __device__ inline int plane_offset(int ilv, int r, int s, int nC) { return ilv ? ilv_offset(ilv, r, s, nC) : r * nC + s; }

enum { EPI_NONE = 0, EPI_XPAY_DOT = 1, EPI_AXPY_NRM_DOT = 2, EPI_RESIDUAL = 3 };

// ---- launchers (tfq_kernels*.hip); all asynchronous on `s` ---------------------------------------
void launch_decide(DevPlan const& d, int phase, hipStream_t s); // phase 0: all, 1: reduce columns only, 2: update ctl only
void launch_probe_decide(DevPlan const& d, int phase, hipStream_t s);

// y[iY] = sum A*X over the pair list, device pointers, native layout (tfqmrgpu_ext.h section 3)
// yOrder (device, optional): a prepared launch order, position i of the launch computes Y block yOrder[i] (tfq_order.cpp)
tfqmrgpuStatus_t launch_multiply(char precision, int lm, int ln, uint32_t nnzbY,
    uint32_t const* starts, uint32_t const* pairs, void const* A, void const* X, void* Y, hipStream_t s, uint32_t const* yOrder = nullptr);
uint32_t multiply_blocks_per_work_group(char precision, int lm, int ln);   // Y blocks per work group of that launch (0: this shape's kernel takes no prepared order)

// layout conversion between the caller's block layout and the native one (tfq_layout.hip)
// direction 0: user -> native (setMatrix), 1: native -> user (getMatrix); one batch of user blocks
// [firstUser, firstUser + nBlocks) whose raw bytes sit in `stage`; u2n: user -> native block index
// ilv: element order of the library-side blocks (see ilv_offset)
// userDbl / nativeDbl: precision of the caller's array and of the library-side array (they differ in the mixed-precision mode only)
void launch_convert(int direction, bool userDbl, bool nativeDbl, void* native, void* stage, uint32_t const* u2n,
    uint32_t firstUser, uint32_t nBlocks, int nR, int nC, int layout, bool trans, bool conj, int ilv, hipStream_t s);
void launch_shadow_hash(DevPlan const& d, hipStream_t s);

} // namespace tfq
