// prepared launch order of the stand-alone multiply (tfq_order.cpp; tfqmrgpuExt_multiplyPrepare)
#pragma once
#include <cstdint>
#include <vector>

namespace tfq {
// a permutation of the nY Y blocks: work group b of the launch computes the blocks out[b * ch .. b * ch + ch - 1] (the last one fewer).
// mode 0: the caller's order; 1: by (column group, row band, column, row) -- the XCDs split the columns; 2: the same, and inside an XCD's part the
// work groups with the most block products first; 3: by (row band, column, row) -- the XCDs split the rows; 4: 1 or 3, whichever operand is
// smaller stays whole per XCD (aBlockElems, xBlockElems: elements of an A block and of an X block).  G: block columns per group.
std::vector<uint32_t> multiply_order(uint32_t nY, uint32_t const* starts, uint32_t const* pairs, uint32_t ch, int mode,
                                     uint32_t aBlockElems, uint32_t xBlockElems, uint32_t G = 4);
} // namespace tfq
