// A prepared launch order for the stand-alone multiply on the CALLER's listing (tfqmrgpuExt_multiplyPrepare, include/tfqmrgpu_ext.h).
//
// The contract of tfqmrgpuExt_multiply is the reference kernel's (tfqmrgpu_blockmult.hxx:9-93): Y blocks in the caller's order, each
// with its run of (A block, X block) pairs.  The reference's own benchmark prepares its launch OUTSIDE the timed loop
// (bench_tfqmrgpu.cu:442-556 reads, sorts and uploads the lists; :289-440 times the multiplications) -- so may a caller of this library:
// which work group computes which Y block is the library's to choose, the listing is not touched.  Host-only code, no device call.
//
// What the order does (the same ideas as the solver's own launch order, tfq_plan.cpp, but found from the pair lists alone -- the
// native interface has no block rows or columns):
//  * block COLUMNS are the connected components of "two Y blocks use the same X block" (an X block of column c is only ever used by
//    Y blocks of column c);
//  * the block ROW of a Y block is ranked by the smallest A index of its pairs (the A blocks of one row are contiguous in a BSR list);
//  * Y blocks are sorted by (group of G columns, band of rows, column, row) and cut into work groups of `ch` blocks: a work group's
//    blocks are neighbouring rows of one column (shared X blocks), neighbouring work groups the same rows of the next column (shared
//    A blocks); the list of work groups is cut into 8 contiguous parts, one per XCD (work groups are dealt round-robin to the XCDs);
//  * mode 3: sorted by (row band, column, row) instead -- the XCDs then split the ROWS, each needs an eighth of A and all of X (mode 1: all of A, an
//    eighth of X); mode 4 chooses between 1 and 3 by which of the two operands is larger;
//  * mode 2: inside an XCD's part the work groups with the most block products start first (a launch of about one round of work
//    groups -- BASELINE config 1 -- then ends with its short ones instead of with whatever the listing put last).
#include "tfq_order.hpp"

#include <algorithm>
#include <numeric>

namespace tfq {

namespace {
struct UnionFind {
    std::vector<uint32_t> p;
    explicit UnionFind(uint32_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
    uint32_t find(uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }
};
} // namespace

std::vector<uint32_t> multiply_order(uint32_t nY, uint32_t const* starts, uint32_t const* pairs, uint32_t ch, int mode, uint32_t aBlockElems, uint32_t xBlockElems, uint32_t G) {
    std::vector<uint32_t> perm(nY);
    std::iota(perm.begin(), perm.end(), 0u);
    if (nY < 2 || 0 == ch || mode <= 0) return perm;
    uint32_t const nPairs = starts[nY];
    uint32_t maxX = 0;
    for (uint32_t q = 0; q < nPairs; ++q) maxX = std::max(maxX, pairs[2 * size_t(q) + 1]);
    // columns: components of the "same X block" relation
    UnionFind uf(nY);
    std::vector<uint32_t> owner(size_t(maxX) + 1, 0xffffffffu);
    std::vector<uint32_t> rowKey(nY, 0xffffffffu);
    for (uint32_t y = 0; y < nY; ++y)
        for (uint32_t q = starts[y]; q < starts[y + 1]; ++q) {
            uint32_t const ia = pairs[2 * size_t(q)], ix = pairs[2 * size_t(q) + 1];
            rowKey[y] = std::min(rowKey[y], ia);
            if (0xffffffffu == owner[ix]) owner[ix] = y; else uf.unite(y, owner[ix]);
        }
    // rank the components by their first Y block, the rows by their smallest A index
    std::vector<uint32_t> compRank(nY, 0xffffffffu), comp(nY);
    uint32_t nComp = 0;
    for (uint32_t y = 0; y < nY; ++y) { uint32_t const r = uf.find(y); if (0xffffffffu == compRank[r]) compRank[r] = nComp++; comp[y] = compRank[r]; }
    std::vector<uint32_t> keys(rowKey);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
    // a band = ch block rows: the distinct row keys are about (rows x columns) many when columns see different A blocks first, so a
    // band is cut by KEY VALUE -- ch rows are about ch x (pairs per Y block) consecutive A indices
    uint32_t const perRow = std::max(1u, (nPairs + nY / 2) / nY);
    uint32_t const bandWidth = std::max(1u, ch * perRow);
    auto band = [&](uint32_t y) { return (0xffffffffu == rowKey[y]) ? 0xffffffffu : rowKey[y] / bandWidth; };
    // mode 4: the library chooses between 1 and 3 by what a launch has to bring into every XCD's L2 -- the XCDs split the COLUMNS (1: every XCD needs
    // all of A and an eighth of X) or the ROWS (3: an eighth of A, all of X): rows when the referenced A blocks outweigh the X blocks
    if (4 == mode) {
        std::vector<uint8_t> seenA;
        uint32_t maxA = 0;
        for (uint32_t q = 0; q < nPairs; ++q) maxA = std::max(maxA, pairs[2 * size_t(q)]);
        seenA.assign(size_t(maxA) + 1, 0);
        size_t nA = 0, nX = 0;
        for (uint32_t q = 0; q < nPairs; ++q) if (!seenA[pairs[2 * size_t(q)]]) { seenA[pairs[2 * size_t(q)]] = 1; ++nA; }
        for (auto o : owner) if (o != 0xffffffffu) ++nX;
        mode = (nA * aBlockElems > nX * xBlockElems) ? 3 : 1;
    }
    if (3 == mode)
        std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
            if (band(a) != band(b)) return band(a) < band(b);
            if (comp[a] != comp[b]) return comp[a] < comp[b];
            return rowKey[a] < rowKey[b];
        });
    else
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) {
        uint32_t const ga = comp[a] / G, gb = comp[b] / G;
        if (ga != gb) return ga < gb;
        if (band(a) != band(b)) return band(a) < band(b);
        if (comp[a] != comp[b]) return comp[a] < comp[b];
        return rowKey[a] < rowKey[b];
    });
    // work groups of ch consecutive entries, eight contiguous parts, dealt round-robin
    uint32_t const nWG = (nY + ch - 1) / ch;
    std::vector<uint32_t> wg(nWG);
    std::iota(wg.begin(), wg.end(), 0u);
    std::vector<uint32_t> begin(9, 0);
    for (uint32_t x = 0; x < 8; ++x) begin[x + 1] = begin[x] + nWG / 8 + (x < nWG % 8 ? 1 : 0);
    if (2 == mode) {
        std::vector<uint32_t> work(nWG, 0);
        for (uint32_t w = 0; w < nWG; ++w)
            for (uint32_t i = w * ch; i < std::min(nY, (w + 1) * ch); ++i) work[w] += starts[perm[i] + 1] - starts[perm[i]];
        for (uint32_t x = 0; x < 8; ++x)
            std::stable_sort(wg.begin() + begin[x], wg.begin() + begin[x + 1], [&](uint32_t a, uint32_t b) { return work[a] > work[b]; });
    }
    // position b of the launch (work group b runs on XCD b % 8) takes work group wg[begin[b % 8] + b / 8]; the last, possibly short work
    // group keeps the last position so that every position before it holds ch blocks
    std::vector<uint32_t> out;
    out.reserve(size_t(nWG) * ch);
    std::vector<uint32_t> slots;
    slots.reserve(nWG);
    for (uint32_t i = 0; i * 8 < nWG + 7; ++i)
        for (uint32_t x = 0; x < 8; ++x)
            if (begin[x] + i < begin[x + 1]) slots.push_back(wg[begin[x] + i]);
    uint32_t const shortWG = (nY % ch) ? nWG - 1 : 0xffffffffu;
    for (uint32_t w : slots) if (w != shortWG) for (uint32_t i = w * ch; i < (w + 1) * ch; ++i) out.push_back(perm[i]);
    if (shortWG != 0xffffffffu) for (uint32_t i = shortWG * ch; i < nY; ++i) out.push_back(perm[i]);
    return out;
}

} // namespace tfq
