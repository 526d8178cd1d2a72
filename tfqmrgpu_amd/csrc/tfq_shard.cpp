// Column sharding for the multi-GPU mode (include/tfqmrgpu_ext.h section 4).
//
// The reference is single-GPU (docs/tfQMRgpu_manual.tex:100-101).  Its data structure makes the
// extension natural: every per-RHS scalar is indexed by the compressed block column
// (tfqmrgpu_linalg.hxx:496,516,646-648), so block columns of X/B are independent linear systems
// that share A.  A rank keeps the X/B blocks of a contiguous range of compressed columns, a
// replica of A and its own pair list; only the stopping test couples the ranks.
#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include "tfq_plan.hpp"

extern "C" tfqmrgpuStatus_t tfqmrgpuExt_shardColumns(int mb,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB,
    int indexOffset, int nranks, int rank, tfqmrgpuShard_t* shard)
{
    using namespace tfq;
    if (!shard || !rowPtrX || !colIndX || !rowPtrB || (nnzbB > 0 && !colIndB)) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (mb < 1 || nnzbX < 1 || nranks < 1 || rank < 0 || rank >= nranks) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    try {   // (no exception may cross the C boundary)
    int const off = indexOffset;
    // compressed columns exactly as createPlan numbers them: ascending original index, empty ones skipped
    std::vector<int32_t> cols(colIndX, colIndX + nnzbX);
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    int const nCols = int(cols.size());
    auto compressed = [&](int32_t c) { return int(std::lower_bound(cols.begin(), cols.end(), c) - cols.begin()); };
    std::vector<int64_t> weight(nCols, 0);
    for (int q = 0; q < nnzbX; ++q) ++weight[compressed(colIndX[q])];
    // contiguous ranges with balanced block counts: rank r owns columns [cut[r], cut[r+1])
    std::vector<int> cut(nranks + 1, nCols);
    cut[0] = 0;
    {
        int64_t const total = nnzbX;
        int64_t run = 0; int r = 1;
        for (int c = 0; c < nCols && r < nranks; ++c) {
            run += weight[c];
            // close range r-1 once it holds its share, but leave at least one column for each later rank
            while (r < nranks && (run * nranks >= total * r || nCols - (c + 1) <= nranks - r)) {
                cut[r] = std::min(c + 1, nCols);
                ++r;
            }
        }
        for (int rr = 1; rr <= nranks; ++rr) cut[rr] = std::max(cut[rr], cut[rr - 1]);
        cut[nranks] = nCols;
    }
    int const c0 = cut[rank], c1 = cut[rank + 1];
    auto mine = [&](int32_t c) { int const k = compressed(c); return k >= c0 && k < c1; };

    std::vector<int32_t> rpX(mb + 1, 0), ciX, xb, rpB(mb + 1, 0), ciB, bb;
    for (int r = 0; r < mb; ++r) {
        for (int32_t q = rowPtrX[r] - off; q < rowPtrX[r + 1] - off; ++q)
            if (mine(colIndX[q])) { ciX.push_back(colIndX[q] - off); xb.push_back(q); }
        rpX[r + 1] = int32_t(ciX.size());
        for (int32_t q = rowPtrB[r] - off; q < rowPtrB[r + 1] - off; ++q)
            if (mine(colIndB[q])) { ciB.push_back(colIndB[q] - off); bb.push_back(q); }
        rpB[r + 1] = int32_t(ciB.size());
    }
    auto dup = [](std::vector<int32_t> const& v) {
        auto p = (int32_t*)std::malloc(std::max<size_t>(1, v.size()) * sizeof(int32_t));
        if (p) std::copy(v.begin(), v.end(), p);
        return p;
    };
    shard->mb = mb; shard->nnzbX = int32_t(ciX.size()); shard->nnzbB = int32_t(ciB.size());
    shard->rowPtrX = dup(rpX); shard->colIndX = dup(ciX); shard->rowPtrB = dup(rpB); shard->colIndB = dup(ciB);
    shard->xBlocks = dup(xb); shard->bBlocks = dup(bb);
    shard->firstCol = c0; shard->nCols = c1 - c0;
    if (!shard->rowPtrX || !shard->colIndX || !shard->rowPtrB || !shard->colIndB || !shard->xBlocks || !shard->bBlocks) {
        tfqmrgpuExt_freeShard(shard);
        return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED);
    }
    return TFQMRGPU_STATUS_SUCCESS;
    } catch (std::bad_alloc const&) { return TFQ_ERR(TFQMRGPU_STATUS_ALLOCATION_FAILED); }
}

extern "C" void tfqmrgpuExt_freeShard(tfqmrgpuShard_t* shard) {
    if (!shard) return;
    std::free(shard->rowPtrX); std::free(shard->colIndX); std::free(shard->rowPtrB);
    std::free(shard->colIndB); std::free(shard->xBlocks); std::free(shard->bBlocks);
    *shard = tfqmrgpuShard_t{};
}
