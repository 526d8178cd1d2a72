// Host-side plan of one bsrsv problem: the reference-visible analysis results
// (bit-exact with real-space/tfQMRgpu createPlan, tfqmrgpu.cu:136-351) plus the
// MI355X-specific device layout derived from them.
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

#include "tfqmrgpu.h"
#include "tfqmrgpu_ext.h"

namespace tfq {

// error code packing (tfqmrgpu.h:179-181 of the reference)
inline tfqmrgpuStatus_t err(int code, int line = 0, int key = 0) {
    return code + TFQMRGPU_CODE_LINE * line + TFQMRGPU_CODE_CHAR * key;
}
#define TFQ_ERR(code) ::tfq::err((code), __LINE__ % 10000)

inline size_t align256(size_t n) { return (n + 255) & ~size_t(255); }

struct Window { size_t offset = 0, bytes = 0; };

// one unit of work of the vector kernels: a run of blocks inside ONE block column
struct ChunkTable {
    std::vector<uint32_t> first;   // [nChunks+1] first internal block of each chunk
    std::vector<uint32_t> col;     // [nChunks]   compressed block column
    std::vector<uint32_t> colPtr;  // [nCols+1]   first chunk of each column
    std::vector<uint32_t> order;   // [nChunks]   launch order of the multiply (XCD aware)
    std::vector<uint32_t> orderB;  // column batches (Plan::colBatch): the chunks of the batches' first columns, in the same kind of order
};

struct Handle {
    void* stream = nullptr;               // hipStream_t
    // multi-GPU stopping test (tfqmrgpu_ext.h section 4)
    void* comm = nullptr;                 // ncclComm_t, RCCL loaded lazily
    int nranks = 1, rank = 0;
    tfqmrgpuReduceMax_t reduceFn = nullptr;
    void* reduceCtx = nullptr;
    double* voteBuf = nullptr;            // 4 doubles of device memory for the collective in front of a solve (RCCL path)
};

struct Plan {
    uint32_t magic = 0x7f51a9d3u;

    // ---- analysis results, identical to the reference's bsrsv_plan_t members ----------
    uint32_t nRows = 0, nCols = 0, nnzbA = 0, nnzbX = 0, nnzbB = 0;
    int indexOffset = 0;
    std::vector<uint32_t> pairs;               // [2*nPairs] (inzA, inzX), user order
    std::vector<uint32_t> starts;              // [nnzbX+1]
    std::vector<uint32_t> subset;              // [nnzbB]
    std::vector<uint16_t> colindx;             // [nnzbX]
    std::vector<int32_t>  original_bsrColIndX; // [nCols]

    // ---- internal (device) order of the X-shaped vectors: sorted by (column, row) --------
    std::vector<uint32_t> col32;       // [nnzbX] compressed column of each user block (32 bit)
    std::vector<uint32_t> rowOfX;      // [nnzbX] block row of each user block
    std::vector<uint32_t> u2i, i2u;    // user index <-> internal index
    std::vector<uint32_t> colStart;    // [nCols+1] internal block range of each column
    std::vector<uint32_t> starts_i;    // [nnzbX+1] pair list regrouped by internal Y index
    std::vector<uint32_t> pairs_i;     // [2*nPairs] (inzA user, inzX internal)
    std::vector<uint32_t> subset_i;    // [nnzbB] internal X index of each B block
    std::vector<uint32_t> bcol;        // [nnzbB] compressed column of each B block
    std::vector<uint32_t> bColPtr, bList; // B blocks grouped by compressed column
    std::vector<uint32_t> bOfX;        // [nnzbX] internal order: B block sitting on this X block, or ~0
    std::vector<uint32_t> rowI;        // [nnzbX] block row, internal order
    ChunkTable chunks;

    // ---- fixed by bufferSize -----------------------------------------------------------------
    int LM = 0, LN = 0;
    char precision = 0;                // 'c', 'z' or 'm' (mixed: float inner iterations, double refinement; tfq_api.hip: run_mixed)
    size_t realBytes = 0;              // 4 or 8 (the storage precision of the iteration vectors: 4 for 'm')
    size_t S = 0;                      // bytes of one X-shaped vector
    size_t bufferBytes = 0;
    bool aOnce = false;                // nPairs <= 1.5 x the A blocks that occur in the pair list: A is streamed once per multiply
    int ilv = 0;                       // element order inside the blocks of this plan's buffer (tfq_device.hpp: ilv_offset)
    int ilvZ = 0;                      // 'm': element order of the double-precision arrays (x, B, A, A x), the one a 'z' plan of this shape has

    // windows into the user's device buffer
    Window wX, wV4, wV5, wV6, wV7, wV8, wV9, wV3, wB, wA;
    Window wRho, wAlfa, wBeta, wC67, wEta, wC67a, wEta2; // [nCols][2][LN] real
    Window wZ, wD, wTau, wVar, wInvBn2;        // double scalars per right-hand side
    Window wStatus;                            // int8 [nCols][LN]
    Window wCtl;                               // device control block (see tfq_solver)
    Window wPz, wPd;                           // per-chunk partial sums (double)
    Window wColRec;                            // per-column stopping-test record [nCols][2] double
    Window wColPart; uint32_t colSegMax = 1;   // shares of the segment work groups of long columns (tfq_colops.hpp: column_total), segments of the longest column
    Window wChunkFirst, wChunkCol, wColChunkPtr, wColStart, wOrigCol, wBofX, wOrder;
    Window wColBatch, wOrderB; std::vector<uint8_t> colBatch;   // batches of block columns with identical row patterns (empty: none); layoutBuffer
    Window wStarts, wPairs, wSubset, wBColPtr, wBList, wU2I, wRowI;
    // 'm' only: the solution, B and A in double; the residual of the refinement as the right-hand side of the inner (float) solve,
    // |b|^2 per right-hand side, the record of the refinement's stopping test
    Window wXz, wBz, wAz, wR, wBn2z, wRefine;
    Window wFold, wSelf;               // arrival counters [nCols + 1] and a device copy of the DevPlan for the folded column operations
    bool selfStale = true;             // the device copy of the DevPlan (wSelf) predates a change of a plan flag (shadow vector, product form): refresh before a folded solve
    bool foldOk = false;               // few enough chunks that an iteration slot is launch latency: fold (tfq_colops.hpp)

    char* buffer = nullptr;            // device buffer registered by setBuffer
    static constexpr int kDepth = 4;   // iterations the host keeps enqueued ahead of the stopping decision
    void* ring = nullptr;              // pinned copies of the control block, one per in-flight iteration
    void* ringEvent[kDepth] = {};      // hipEvent_t behind each copy
    int shadowMode = TFQMRGPU_SHADOW_HASH;
    bool v3IsHash = false;             // the buffer's v3 holds the counter-based hash (set by setBuffer, cleared by a user-supplied vector)
    bool haveB = false;
    bool threeProducts = false;        // tfqmrgpuExt_setThreeProductMultiply: Gauss' three real products per complex one in the double multiplies
    std::vector<double> cycleResidual; // 'm': relative residual (double arithmetic) in front of every inner solve and at the end
    std::vector<int32_t> cycleIterations; // 'm': float iterations of every inner solve
    int refinementCycles = 0;
    double mixedFloor = 0;             // 'm': the gain at which the first float solve of this plan's last solve ran into its floor (0: not known; forgotten with a new A)
    // user-defined operator (tfqmrgpu_ext.h section 5): callback, and library-owned device scratch
    // [xu | yu | i2u | colindx in the caller's block order]
    void* opFn = nullptr; void* opCtx = nullptr;
    char* opScratch = nullptr;

    // ---- results of the last solve -------------------------------------------------------------
    double residuum_reached = 0, flops_performed = -1, flops_performed_all = 0;
    int iterations_needed = -1;
    std::vector<double> boundHistory;
    int profiling = 0;                 // 0 off, 1 every kernel class, 2 the two fused multiplies only
    int64_t profLaunches[16] = {};
    double profMs[16] = {};
    int64_t profGatedLaunches[16] = {};   // launches that found the solve stopped / no probe requested
    double profGatedMs[16] = {};
    int64_t profFirstLaunches[16] = {};   // of profLaunches: the launches of the first iteration (they skip the operands that are zero there)
    double profFirstMs[16] = {};

    size_t nPairs() const { return pairs.size() / 2; }
};

inline Plan* asPlan(tfqmrgpuBsrsvPlan_t p) {
    auto q = reinterpret_cast<Plan*>(p);
    return (q && q->magic == 0x7f51a9d3u) ? q : nullptr;
}

// index analysis, tfq_plan.cpp
tfqmrgpuStatus_t analyse(Plan& p, int mb,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB,
    int indexOffset, int echo);

// derive chunk tables + buffer windows for (LM, LN, precision); tfq_plan.cpp
tfqmrgpuStatus_t layoutBuffer(Plan& p, int LM, int LN, char precision);

// the 15 compiled (ldA, ldB) pairs of the reference (allowed_block_sizes.h:4-18)
extern int const kAllowedBlockSizes[15][2];
bool blockSizeAllowed(int lm, int ln);

} // namespace tfq
