// Index analysis of A*X==B (host only) and the device buffer layout.
//
// analyse() restates what the reference does in tfqmrgpu_bsrsv_createPlan
// (real-space/tfQMRgpu tfQMRgpu/source/tfqmrgpu.cu:161-339): the results `pairs`, `starts`,
// `subset`, `colindx`, `original_bsrColIndX`, `nCols` are bit-identical, including the order
// of the pair list (block row, then X-block order inside the row, then A-block order inside
// the row) and the "first match wins" rule of its linear search (bsr.hxx:27-39).
// The search itself is done with one sorted (column, position) list per block row, so the
// cost is O(nPairs log) instead of the reference's O(nnzbX * nnz/rowA * nnz/rowX).
#include "tfq_plan.hpp"
#include "tfq_switch.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <numeric>

namespace tfq {

int const kAllowedBlockSizes[15][2] = {
    { 4, 4}, { 4, 5}, { 4, 8}, { 4,32}, { 8, 8}, { 8, 9}, { 8,10}, { 8,32}, { 8,64},
    {16,16}, {16,32}, {16,64}, {32,32}, {32,64}, {64,64}};

bool blockSizeAllowed(int lm, int ln) {
    for (auto const& s : kAllowedBlockSizes) if (s[0] == lm && s[1] == ln) return true;
    return false;
}

namespace {

// per block row: (raw column index, position) sorted, to find the FIRST block with a given column
struct RowLookup {
    std::vector<std::pair<int32_t, int32_t>> key; // sorted by (col, pos)
    std::vector<int64_t> rowBegin;                // [mb+1]
    void build(int mb, int32_t const* rowPtr, int32_t const* colInd, int off) {
        rowBegin.resize(mb + 1);
        int64_t n = 0;
        for (int r = 0; r < mb; ++r) { rowBegin[r] = n; n += std::max(0, rowPtr[r + 1] - rowPtr[r]); }
        rowBegin[mb] = n;
        key.resize(n);
        for (int r = 0; r < mb; ++r) {
            auto w = rowBegin[r];
            for (int32_t q = rowPtr[r] - off; q < rowPtr[r + 1] - off; ++q) key[w++] = {colInd[q], q};
            std::sort(key.begin() + rowBegin[r], key.begin() + rowBegin[r + 1]);
        }
    }
    // first position in row r whose raw column equals `col`, or -1
    int32_t find(int r, int32_t col) const {
        auto b = key.begin() + rowBegin[r], e = key.begin() + rowBegin[r + 1];
        auto it = std::lower_bound(b, e, std::make_pair(col, int32_t(-2147483647 - 1)));
        return (it != e && it->first == col) ? it->second : -1;
    }
};

} // namespace

tfqmrgpuStatus_t analyse(Plan& p, int mb,
    int32_t const* rowPtrA, int nnzbA, int32_t const* colIndA,
    int32_t const* rowPtrX, int nnzbX, int32_t const* colIndX,
    int32_t const* rowPtrB, int nnzbB, int32_t const* colIndB,
    int indexOffset, int echo)
{
    // plausibility checks, same order as the reference (tfqmrgpu.cu:166-172)
    if (mb < 1)        return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (nnzbX < 1)     return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (nnzbB > nnzbX) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (int64_t(nnzbA) > int64_t(mb) * mb) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (!rowPtrA || !rowPtrX || !rowPtrB) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);
    if (nnzbA != rowPtrA[mb] - rowPtrA[0]) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (nnzbX != rowPtrX[mb] - rowPtrX[0]) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if (nnzbB != rowPtrB[mb] - rowPtrB[0]) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    if ((nnzbA > 0 && !colIndA) || !colIndX || (nnzbB > 0 && !colIndB)) return TFQ_ERR(TFQMRGPU_POINTER_INVALID);

    int const off = indexOffset;
    // the reference trusts its inputs beyond this point; a library that launches device kernels
    // from these lists must not, so malformed row pointers / columns of A are rejected here
    for (int r = 0; r < mb; ++r) {
        if (rowPtrA[r + 1] < rowPtrA[r] || rowPtrX[r + 1] < rowPtrX[r] || rowPtrB[r + 1] < rowPtrB[r])
            return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    }
    if (rowPtrA[0] != off || rowPtrX[0] != off || rowPtrB[0] != off) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    for (int q = 0; q < nnzbA; ++q) {
        auto const k = colIndA[q] - off;
        if (k < 0 || k >= mb) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    }

    p.nRows = mb; p.nnzbA = nnzbA; p.nnzbX = nnzbX; p.nnzbB = nnzbB;

    RowLookup look;
    look.build(mb, rowPtrX, colIndX, off);

    // ---- pair list of Y = A*X, Y has the pattern of X (tfqmrgpu.cu:183-230) ----------------
    p.starts.assign(size_t(nnzbX) + 1, 0);
    p.pairs.clear();
    p.pairs.reserve(2 * (size_t(nnzbX) * size_t(nnzbA) / size_t(mb) + 1));
    p.rowOfX.resize(nnzbX);
    for (int irow = 0; irow < mb; ++irow) {
        for (int32_t inzy = rowPtrX[irow] - off; inzy < rowPtrX[irow + 1] - off; ++inzy) {
            p.rowOfX[inzy] = irow;
            auto const jcol = colIndX[inzy]; // raw (offset included), compared with raw values
            for (int32_t inza = rowPtrA[irow] - off; inza < rowPtrA[irow + 1] - off; ++inza) {
                auto const krow = colIndA[inza] - off;
                auto const inzx = look.find(krow, jcol);
                if (inzx >= 0) { p.pairs.push_back(uint32_t(inza)); p.pairs.push_back(uint32_t(inzx)); }
            }
            p.starts[size_t(inzy) + 1] = uint32_t(p.pairs.size() / 2);
        }
    }
    p.pairs.shrink_to_fit();
    {   // how often a multiply uses an A block on average (decides the cache policy of the A operands, tfq_spmm.hip)
        std::vector<uint8_t> used(size_t(nnzbA), 0);
        size_t distinctA = 0;
        for (size_t q = 0; q < p.pairs.size(); q += 2) if (!used[p.pairs[q]]) { used[p.pairs[q]] = 1; ++distinctA; }
        p.aOnce = (distinctA > 0) && (2 * p.nPairs() <= 3 * distinctA);
    }
    if (echo > 6) std::printf("# tfqmrgpu_bsrsv_createPlan: found %zu pairs in A*X multiplication\n", p.nPairs());

    // ---- B must live on blocks of X (tfqmrgpu.cu:233-251) ---------------------------------------
    p.subset.assign(nnzbB, 0);
    for (int irow = 0; irow < mb; ++irow) {
        for (int32_t inzb = rowPtrB[irow] - off; inzb < rowPtrB[irow + 1] - off; ++inzb) {
            auto const inzx = look.find(irow, colIndB[inzb]);
            if (inzx < 0) {
                if (echo > 0) std::printf("# tfqmrgpu_bsrsv_createPlan: in row #%d B has col #%d but X does not!\n",
                                          irow + off, colIndB[inzb]);
                return err(TFQMRGPU_B_IS_NOT_SUBSET_OF_X, irow); // the row travels in the line field
            }
            p.subset[inzb] = uint32_t(inzx);
        }
    }

    // ---- compress the block columns of X (tfqmrgpu.cu:254-315) ------------------------------
    // The reference histograms the range [min, max] of the column indices and numbers the non-empty columns in ascending
    // order; the same numbering from a sorted list of the distinct indices, so that one stray index cannot ask for 2^32 counters.
    std::vector<int32_t> distinct(colIndX, colIndX + nnzbX);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    int32_t const lo = distinct.front(), hi = distinct.back();
    int64_t const nc = int64_t(hi) - lo + 1;
    if (echo > 5) std::printf("# tfqmrgpu_bsrsv_createPlan: column indices of X are in [%d, %d]\n", lo, hi);
    uint32_t const nb = uint32_t(distinct.size());
    int64_t const nempty = nc - int64_t(nb);
    if (echo > 5) std::printf("# tfqmrgpu_bsrsv_createPlan: found %lld empty columns and %u columns with entries\n", (long long)nempty, nb);
    if (nempty > 0 && echo > 0) std::printf("# tfqmrgpu_bsrsv_createPlan: found %lld empty columns in X!\n", (long long)nempty);
    if (nb < 1) return TFQ_ERR(TFQMRGPU_UNDOCUMENTED_ERROR);
    // the compressed column index is 16 bit in the reference (colIndex_t, tfqmrgpu.hxx:59; asserted at tfqmrgpu_core.hxx:81)
    // and in this library's plan view and operator callback: more block columns are refused, not silently truncated
    if (nb > 65536u) return TFQ_ERR(TFQMRGPU_NO_IMPLEMENTATION);
    p.nCols = nb;
    p.colindx.resize(nnzbX);
    p.col32.resize(nnzbX);
    p.original_bsrColIndX.assign(nb, 0);
    for (int q = 0; q < nnzbX; ++q) {
        auto const jb = uint32_t(std::lower_bound(distinct.begin(), distinct.end(), colIndX[q]) - distinct.begin());
        p.original_bsrColIndX[jb] = colIndX[q];
        p.colindx[q] = uint16_t(jb);   // the reference's colIndex_t is 16 bit (tfqmrgpu.hxx:59)
        p.col32[q] = jb;
    }

    // ---- every column of X needs a block of B (tfqmrgpu.cu:316-337) ---------------------------
    {
        std::vector<uint32_t> bPerCol(nb, 0);
        for (int q = 0; q < nnzbB; ++q) ++bPerCol[p.col32[p.subset[q]]];
        uint32_t nzero = 0;
        for (uint32_t jb = 0; jb < nb; ++jb) nzero += (bPerCol[jb] < 1);
        if (nzero > 0) {
            if (echo > 0) std::printf("# tfqmrgpu_bsrsv_createPlan: found %u zero columns in B!\n", nzero);
            return err(TFQMRGPU_B_HAS_A_ZERO_COLUMN, int(nzero)); // the count travels in the line field
        }
    }

    // ==== MI355X device order: X-shaped vectors sorted by (compressed column, block row) =========
    // Every vector kernel then works on runs of blocks that share one column (uniform per-RHS
    // scalars, register-level reductions), and a shard of columns is one contiguous range.
    p.i2u.resize(nnzbX);
    std::iota(p.i2u.begin(), p.i2u.end(), 0u);
    std::stable_sort(p.i2u.begin(), p.i2u.end(), [&](uint32_t a, uint32_t b) {
        if (p.col32[a] != p.col32[b]) return p.col32[a] < p.col32[b];
        return p.rowOfX[a] < p.rowOfX[b];
    });
    p.u2i.resize(nnzbX);
    for (uint32_t i = 0; i < uint32_t(nnzbX); ++i) p.u2i[p.i2u[i]] = i;
    p.colStart.assign(size_t(nb) + 1, 0);
    for (int q = 0; q < nnzbX; ++q) ++p.colStart[p.col32[q] + 1];
    for (uint32_t jb = 0; jb < nb; ++jb) p.colStart[jb + 1] += p.colStart[jb];

    p.starts_i.assign(size_t(nnzbX) + 1, 0);
    p.pairs_i.resize(p.pairs.size());
    {
        size_t w = 0;
        for (uint32_t i = 0; i < uint32_t(nnzbX); ++i) {
            auto const u = p.i2u[i];
            for (auto q = p.starts[u]; q < p.starts[u + 1]; ++q) {
                p.pairs_i[2 * w]     = p.pairs[2 * size_t(q)];
                p.pairs_i[2 * w + 1] = p.u2i[p.pairs[2 * size_t(q) + 1]];
                ++w;
            }
            p.starts_i[i + 1] = uint32_t(w);
        }
    }
    p.subset_i.resize(nnzbB);
    p.bcol.resize(nnzbB);
    {
        std::vector<uint8_t> seen(nnzbX, 0);
        for (int q = 0; q < nnzbB; ++q) {
            p.subset_i[q] = p.u2i[p.subset[q]];
            p.bcol[q] = p.col32[p.subset[q]];
            // two blocks of B on the same block of X would make the scatter ambiguous on a GPU
            if (seen[p.subset[q]]) return TFQ_ERR(TFQMRGPU_NO_IMPLEMENTATION);
            seen[p.subset[q]] = 1;
        }
    }
    p.bOfX.assign(nnzbX, 0xffffffffu);
    for (int q = 0; q < nnzbB; ++q) p.bOfX[p.subset_i[q]] = uint32_t(q);
    p.bColPtr.assign(size_t(nb) + 1, 0);
    for (int q = 0; q < nnzbB; ++q) ++p.bColPtr[p.bcol[q] + 1];
    for (uint32_t jb = 0; jb < nb; ++jb) p.bColPtr[jb + 1] += p.bColPtr[jb];
    p.bList.resize(nnzbB);
    {
        std::vector<uint32_t> fill(p.bColPtr.begin(), p.bColPtr.end() - 1);
        for (int q = 0; q < nnzbB; ++q) p.bList[fill[p.bcol[q]]++] = uint32_t(q);
    }
    p.rowI.resize(nnzbX);
    for (uint32_t i = 0; i < uint32_t(nnzbX); ++i) p.rowI[i] = p.rowOfX[p.i2u[i]];
    return TFQMRGPU_STATUS_SUCCESS;
}

// ---------------------------------------------------------------------------------------------
// Device buffer layout.  The reference carves its buffer in tfqmrgpu_core.hxx:49-84 and
// tfqmrgpu_blocksparse.hxx:46-51 (X first, 256-byte granules, tfqmrgpu_util.hxx:56-82); the
// layout here is this library's own: no 2^ceil(log2 nnzbX) reduction scratch (zvv/dvv), instead
// one small partial-sum record per chunk.
tfqmrgpuStatus_t layoutBuffer(Plan& p, int LM, int LN, char precision) {
    // 'm' (mixed precision): the iteration runs on a complex<float> plan; x, B, A are kept in double next to it (below)
    bool const mixed = ('m' == precision);
    char const iterPrec = mixed ? 'c' : precision;
    p.LM = LM; p.LN = LN; p.precision = precision;
    p.realBytes = ('z' == iterPrec) ? 8 : 4;
    size_t const blockElems = size_t(2) * LM * LN;
    p.S = size_t(p.nnzbX) * blockElems * p.realBytes;
    // groups of rows interleaved (16-byte accesses for one column) where a multiply kernel is written for it: 16 x 16 and 8 x 8
    // complex<double> (pairs), 16 x 16, 16 x 32 and 32 x 32 complex<float> (quads).  A lab build (-DTFQ_LAB) can keep the native order
    // with TFQMRGPU_ILV=0 (A/B runs), =16: only 16 x 16 z, =2: only the double shapes, =3: all but the float shapes beyond 16 x 16
    int const ilvEnv = lab_switch("TFQMRGPU_ILV", 1);
    auto const ilvOf = [&](char prec) {
        int ilv = 0;
        if (ilvEnv && 'z' == prec && ((16 == LM && 16 == LN) || (8 == LM && 8 == LN && ilvEnv != 16))) ilv = 2;
        if ((1 == ilvEnv || 2 == ilvEnv) && 'z' == prec && 8 == LM && (32 == LN || 64 == LN)) ilv = 2;      // k_spmm_ilv8w (r03)
        if (1 == ilvEnv && 'z' == prec && 8 == LM && (9 == LN || 10 == LN) && lab_switch("TFQMRGPU_ILV89", 1)) ilv = 2;   // k_spmm_ilv8w with a ragged second column group (r04)
        if ((1 == ilvEnv || 3 == ilvEnv) && 'c' == prec && 16 == LM && 16 == LN) ilv = 4;
        if (1 == ilvEnv && 'c' == prec && (16 == LM || 32 == LM) && 32 == LN) ilv = 4;   // 16 x 32, 32 x 32 (k_spmm_ilvf)
        if (1 == ilvEnv && 'c' == prec && 8 == LM && (8 == LN || 32 == LN || 64 == LN)) ilv = 4;   // k_spmm_ilv8f (r03)
        if (1 == ilvEnv && 'c' == prec && LM >= 16 && 64 == LN && lab_switch("TFQMRGPU_ILV64", 1)) ilv = 4;   // k_spmm_ilvf on column halves (r03)
        return ilv;
    };
    p.ilv = ilvOf(iterPrec);
    p.ilvZ = mixed ? ilvOf('z') : 0;

    // Column batches (k_spmm_ilv8b, 8 x 8 complex<double>): runs of up to kColBatchMax neighbouring block columns whose row patterns are identical -- same
    // block rows in the same order, hence the same chunk cuts and the same A blocks per row -- are multiplied together, one A fetch for the block products of all of them.  Dense right-hand-side columns (BASELINE config 5) qualify; columns truncated around different centres do not.  Lab: TFQMRGPU_BATCH=1 off.
    p.colBatch.clear();
    if ('z' == iterPrec && 8 == LM && 8 == LN && 2 == p.ilv) {
        int const maxB = std::min(kColBatchMax, std::max(1, lab_switch("TFQMRGPU_BATCH", kColBatchMax)));
        p.colBatch.assign(p.nCols, uint8_t(1 << 4));
        bool any = false;
        for (uint32_t c = 0; c < p.nCols; ) {
            uint32_t const n0 = p.colStart[c + 1] - p.colStart[c];
            uint32_t nb = 1;
            while (int(nb) < maxB && c + nb < p.nCols && p.colStart[c + nb + 1] - p.colStart[c + nb] == n0 && n0 > 0 &&
                   std::equal(p.rowI.begin() + p.colStart[c], p.rowI.begin() + p.colStart[c + 1], p.rowI.begin() + p.colStart[c + nb])) ++nb;
            for (uint32_t k = 0; k < nb; ++k) p.colBatch[c + k] = uint8_t((nb << 4) | k);
            any = any || nb > 1;
            c += nb;
        }
        if (!any) p.colBatch.clear();
    }

    // chunks: runs of CH blocks inside one column, sized so that a chunk of one vector is 8..16 KiB and the
    // grid has a few thousand work groups when the problem is large enough
    {
        size_t const blockBytes = blockElems * p.realBytes;
        size_t target = p.S / 4096;
        size_t maxKiB = 16;   // P2: 64 -> 16 KiB takes the vector kernels from 4.2 to 17 rounds of work groups (tail 16 % -> 1 %): iteration 2.80 -> 2.69 ms
        maxKiB = size_t(std::max(8, lab_switch("TFQMRGPU_CHUNK_KIB", 16)));
        target = std::min<size_t>(std::max<size_t>(target, 8 * 1024), maxKiB * 1024);
        uint32_t CH = uint32_t(std::max<size_t>(1, target / blockBytes));
        // every wave of a work group wants a unit of work: the MFMA multiply cuts a block into strips of 16 or
        // 32 rows (RowTiles in tfq_spmm.hip), one strip per wave and pass
        if (LM % 16 == 0 && LN % 16 == 0) {
            int const mt = LM / 16, ms = (mt % 2 == 0 && 2 * (LN / 16) * p.realBytes <= 32) ? 2 : 1;
            uint32_t const perBlock = uint32_t(mt / ms);
            CH = std::max(CH, (4 + perBlock - 1) / perBlock);
        }
        auto& c = p.chunks;
        c.first.clear(); c.col.clear(); c.colPtr.assign(size_t(p.nCols) + 1, 0);
        std::vector<uint32_t> bandOf;                        // band of CH block rows in which a chunk starts
        for (uint32_t jb = 0; jb < p.nCols; ++jb) {
            c.colPtr[jb] = uint32_t(c.col.size());
            for (uint32_t b = p.colStart[jb]; b < p.colStart[jb + 1]; b += CH) {
                c.first.push_back(b); c.col.push_back(jb); bandOf.push_back(p.rowI[b] / CH);
            }
        }
        c.colPtr[p.nCols] = uint32_t(c.col.size());
        c.first.push_back(p.nnzbX);

        // Launch order of the multiply.  Work groups are dealt round-robin to the 8 XCDs (observed; used for
        // speed only, never for correctness), each XCD has its own 4 MiB L2.  Chunks are sorted by
        // (group of G block columns, band of block rows, column) and the sorted list is cut into 8
        // contiguous parts, one per XCD: inside an XCD consecutive work groups walk down the row bands of G
        // columns together, so the A blocks of a band are re-read from that L2 (measured on P2: fused
        // multiply 0.92 -> 0.82 ms; larger G loses the X blocks shared by neighbouring bands).
        // (Tried and dropped in round 1: one chunk per WAVE with the 4 waves of a work group on 4 columns
        //  of the same band -- 20 % slower; with all A traffic removed artificially the kernel only
        //  reaches 0.77 ms, so A re-reads are not what bounds it any more.)
        auto envu = [](char const* name, uint32_t dflt) { return uint32_t(lab_switch(name, int(dflt))); };
        uint32_t const n = uint32_t(c.col.size());
        uint32_t const mode = envu("TFQMRGPU_ORDER", 1), G = std::max(1u, envu("TFQMRGPU_ORDER_G", 4));   // 8 until the epilogue streams went non-temporal / the shadow vector stopped being read: now 4 (P2 iteration 2.649 -> 2.626 ms, 8x8 z 2.819 -> 2.792, 2: 2.655)
        uint32_t const BM = std::max(1u, envu("TFQMRGPU_ORDER_BANDMULT", 1));   // bands of BM*CH block rows
        std::vector<uint32_t> sorted(n);
        std::iota(sorted.begin(), sorted.end(), 0u);
        c.order = sorted;
        if (mode && n >= 64) {
            std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b) {
                uint32_t const ga = c.col[a] / G, gb = c.col[b] / G;
                if (ga != gb) return ga < gb;
                if (bandOf[a] / BM != bandOf[b] / BM) return bandOf[a] / BM < bandOf[b] / BM;
                return c.col[a] < c.col[b];
            });
            uint32_t const q = n / 8, r = n % 8;
            std::vector<uint32_t> begin(9, 0);
            for (uint32_t x = 0; x < 8; ++x) begin[x + 1] = begin[x] + q + (x < r ? 1 : 0);
            uint32_t w = 0;
            for (uint32_t i = 0; i <= q; ++i)
                for (uint32_t x = 0; x < 8; ++x)
                    if (begin[x] + i < begin[x + 1]) c.order[w++] = sorted[begin[x] + i];
        }
        c.orderB.clear();
        if (!p.colBatch.empty()) {   // the same order over the chunks of the batches' first columns only: every work group of that launch has work
            std::vector<uint32_t> lead;
            for (uint32_t ch : sorted) if (0 == (p.colBatch[c.col[ch]] & 15)) lead.push_back(ch);
            uint32_t const nl = uint32_t(lead.size()), q = nl / 8, r = nl % 8;
            c.orderB = lead;
            if (mode && n >= 64) {
                std::vector<uint32_t> begin(9, 0);
                for (uint32_t x = 0; x < 8; ++x) begin[x + 1] = begin[x] + q + (x < r ? 1 : 0);
                uint32_t w = 0;
                for (uint32_t i = 0; i <= q; ++i)
                    for (uint32_t x = 0; x < 8; ++x)
                        if (begin[x] + i < begin[x + 1]) c.orderB[w++] = lead[begin[x] + i];
            }
        }
    }
    size_t const nChunks = p.chunks.col.size();
    if (nChunks <= size_t(lab_switch("TFQMRGPU_FOLD_MAX", kFoldMax))) { p.colBatch.clear(); p.chunks.orderB.clear(); }   // small plans fold their column operations into the multiplies' tails instead (below)

    size_t at = 0;
    auto take = [&](Window& w, size_t bytes) { w.offset = at; w.bytes = bytes; at = align256(at + bytes); };
    // (r03: X-shaped vectors that lie exactly 2^k bytes apart -- config 4: 2 GiB, config 5: 512 MiB -- do NOT alias on the memory channels:
    //  a gap of 4 KiB ... 1 MiB behind each vector changed no kernel time, profiles/r03_ab_ilv8w.txt)
    take(p.wX, p.S);                                  // the solution stays first, as in the reference
    take(p.wV4, p.S); take(p.wV5, p.S); take(p.wV6, p.S);
    take(p.wV7, p.S); take(p.wV8, p.S); take(p.wV9, p.S);
    take(p.wV3, size_t(p.nnzbX) * blockElems * sizeof(float)); // the shadow vector is float for 'z' too
    take(p.wB, mixed ? 0 : size_t(p.nnzbB) * blockElems * p.realBytes);   // ('m' keeps B in double only, wBz)
    size_t const cs = size_t(p.nCols) * 2 * LN * p.realBytes;
    take(p.wRho, cs); take(p.wAlfa, cs); take(p.wBeta, cs); take(p.wC67, cs); take(p.wEta, cs);
    take(p.wC67a, cs); take(p.wEta2, cs);
    take(p.wZ,   size_t(p.nCols) * 2 * LN * sizeof(double));
    take(p.wD,   size_t(p.nCols) * LN * sizeof(double));
    take(p.wTau, size_t(p.nCols) * LN * sizeof(double));
    take(p.wVar, size_t(p.nCols) * LN * sizeof(double));
    take(p.wInvBn2, size_t(p.nCols) * LN * sizeof(double));
    take(p.wStatus, size_t(p.nCols) * LN);
    take(p.wCtl, 4096);
    take(p.wPz, nChunks * 2 * LN * sizeof(double));
    take(p.wPd, nChunks * LN * sizeof(double));
    take(p.wColRec, size_t(p.nCols) * 2 * sizeof(double)); // per-column stopping-test record
    {   // long columns are summed by several work groups (tfq_colops.hpp: col_segments, column_total; the same two formulas)
        uint32_t const segLen = uint32_t(256 / LN) * 16u;
        p.colSegMax = 1;
        for (uint32_t jb = 0; jb < p.nCols; ++jb) {
            uint32_t const n = p.chunks.colPtr[jb + 1] - p.chunks.colPtr[jb];
            p.colSegMax = std::max(p.colSegMax, (n <= 4 * segLen) ? 1u : (n + segLen - 1) / segLen);
        }
        uint32_t const K = 64;             // kColSlot
        take(p.wColPart, (nChunks / K + p.nCols + 2) * 3 * size_t(LN) * sizeof(double));
    }
    take(p.wChunkFirst, (nChunks + 1) * sizeof(uint32_t));
    take(p.wChunkCol, nChunks * sizeof(uint32_t));
    take(p.wOrder, nChunks * sizeof(uint32_t));
    take(p.wColChunkPtr, (size_t(p.nCols) + 1) * sizeof(uint32_t));
    take(p.wColStart, (size_t(p.nCols) + 1) * sizeof(uint32_t));
    take(p.wOrigCol, size_t(p.nCols) * sizeof(int32_t));
    take(p.wColBatch, size_t(p.nCols));
    take(p.wOrderB, p.chunks.orderB.size() * sizeof(uint32_t));
    take(p.wBofX, size_t(p.nnzbX) * sizeof(uint32_t));         // B block on each X block or ~0
    take(p.wStarts, p.starts_i.size() * sizeof(uint32_t));
    take(p.wPairs, p.pairs_i.size() * sizeof(uint32_t));
    take(p.wSubset, size_t(p.nnzbB) * sizeof(uint32_t));
    take(p.wBColPtr, (size_t(p.nCols) + 1) * sizeof(uint32_t));
    take(p.wBList, size_t(p.nnzbB) * sizeof(uint32_t));
    take(p.wU2I, size_t(p.nnzbX) * sizeof(uint32_t));
    take(p.wRowI, size_t(p.nnzbX) * sizeof(uint32_t));
    take(p.wFold, (size_t(p.nCols) + 1) * sizeof(uint32_t));
    take(p.wSelf, 1024);
    // Small systems fold the column operations into the producers' tails (tfq_colops.hpp): six launches less per iteration slot.
    // "Small" = at most kFoldMax = 384 chunks, i.e. work groups per multiply: measured with one build and the switch (scripts/fold_crossover.py).
    // r03, arrivals with fences: gains below 128 chunks, LOSES 10-14 % at 256 ... 288.  r04, fence-free arrivals (co_store / co_load): 22 chunks -9 %,
    // 256-288 chunks -6 ... -9.5 %, 576 chunks +1.6 %, 1024 and more +6 ... +27 % (every work group waits for its own stores in front of the arrival):
    // profiles/r04_small_systems.txt.  Lab builds: TFQMRGPU_FOLD_MAX chunks.
    p.foldOk = (nChunks <= size_t(lab_switch("TFQMRGPU_FOLD_MAX", kFoldMax)));
    take(p.wA, size_t(p.nnzbA) * 2 * LM * LM * p.realBytes);
    if (mixed) {
        // Mixed precision: float vectors for the iteration (above), and in double the solution, B, A (the refinement's residual
        // b - A x is computed in double) and |b|^2.  The product A x of the refinement lives in the work vectors v4 ... v7, which are
        // free between two inner solves (4 float vectors = 2 double ones).  All in all 11 float-sized vectors (x, v4 ... v9, v3, R, and
        // x in double) against 15 for a 'z' plan.
        take(p.wXz, size_t(p.nnzbX) * blockElems * sizeof(double));
        take(p.wR, p.S);
        take(p.wBz, size_t(p.nnzbB) * blockElems * sizeof(double));
        take(p.wAz, size_t(p.nnzbA) * 2 * LM * LM * sizeof(double));
        take(p.wBn2z, size_t(p.nCols) * LN * sizeof(double));
        take(p.wRefine, 256);
    }
    p.bufferBytes = at + 256;
    return TFQMRGPU_STATUS_SUCCESS;
}

} // namespace tfq
