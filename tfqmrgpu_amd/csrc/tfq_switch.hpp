// Tuning switches.  The product's dispatch is FROZEN: every switch has the value that was measured best and
// nothing in libtfQMRgpu.so reads the environment to choose a kernel or a layout.  A lab build (libtfQMRgpu_lab.so, the same
// sources compiled with -DTFQ_LAB by the Makefile) reads TFQMRGPU_* from the environment instead, for A/B runs of whole builds and
// for the tests that drive every non-default code path (tests/test_gpu_hash_mode.py).
#pragma once
#include <cstdlib>

namespace tfq {
// block columns multiplied together at most (tfq_plan.cpp: Plan::colBatch, tfq_spmm.hip: k_spmm_ilv8b; with 4 hipcc does not fit the registers of
// three waves per SIMD without scratch)
constexpr int kColBatchMax = 2;
// plans of at most this many chunks fold their column operations into the producers' tails (tfq_colops.hpp; r01-r03: 128 with fenced arrivals; r04: the
// fence-free form gains 7-10 % up to 261 chunks and loses 2 % at 576, profiles/r04_small_systems.txt)
constexpr int kFoldMax = 384;

inline int lab_switch(char const* name, int dflt) {
#ifdef TFQ_LAB
    auto const v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}
} // namespace tfq
