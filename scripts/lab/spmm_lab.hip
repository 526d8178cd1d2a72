// Kernel laboratory for the fused 16x16 complex<double> multiply: runs kernel variants on the index lists written by
// scripts/lab/make_plan.py, times them with HIP events and checks them against the shipped kernel on the same lists.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude -Itfqmrgpu_amd/csrc scripts/lab/spmm_lab.hip -o scripts/bin/spmm_lab
// usage: spmm_lab <plan dir> <epi 1|2> <variant> [reps]
#include "../../tfqmrgpu_amd/csrc/tfq_spmm.hip"
// the lab kernels keep the one-hash-per-real shadow vector they were measured with (the product moved to shadow_quad)
namespace tfq { __host__ __device__ inline float shadow_value(uint64_t key, uint32_t e) {
    uint64_t const h = splitmix64(key + uint64_t(e) * 0xd1342543de82ef95ull);
    return float((h >> 40) + 1) * (1.f / 16777216.f);
} }

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

namespace tfq {

// LDS-DMA hidden from the compiler's wait-count bookkeeping (it would otherwise put vmcnt(0) in front of every later LDS
// read): 16 bytes per lane from `gsrc` (per lane) to LDS byte address `ldsDst` + 16 * lane (wave-uniform).  The caller
// waits with its own s_waitcnt vmcnt + barrier before reading.
__device__ inline void glds16(void const* gsrc, uint32_t ldsDst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
}
__device__ inline uint32_t lds_addr(void const* p) { return uint32_t(size_t((__attribute__((address_space(3))) void const*)p)); }
#define GPTR(p) ((__attribute__((address_space(1))) void const*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---------------------------------------------------------------------------------------------------
// V2: the X blocks of the chunk (= the Y pattern's own blocks: contiguous in memory) are staged ONCE in LDS by LDS-DMA;
// pairs whose X block lies inside the chunk (listed first, nInside[y] of them) read it from there, the others from
// global memory as before.  A operands always come from global memory.
template <int EPI, bool HASH, int CHMAX, int NW, int ABL>
__global__ __launch_bounds__(NW * 64, (2 * NW) / 4) void k_spmm_ldsx(SpmmArgs a, uint32_t const* __restrict__ nInside) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    __shared__ double lds[CHMAX * 2 * P + NW * 3 * LN];
    double* const red = lds + CHMAX * 2 * P;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;

    // stage: wave w copies blocks w, w + 4, ...; one block = 4 KiB = 4 wave-wide 16-byte DMAs
    for (uint32_t b = wave; b < nb; b += NW) {
        double const* src = (double const*)a.X + size_t(first + b) * 2 * P;
#pragma unroll
        for (int piece = 0; piece < 4; ++piece)
            __builtin_amdgcn_global_load_lds(GPTR(src + piece * 128 + lane * 2), LPTR(lds + b * 2 * P + piece * 128), 16, 0, 0);
    }

    R sr[1], si[1];
    sr[0] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc];
    si[0] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1][1] = {};

    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMAs have landed (a wait the compiler's own bookkeeping sees)
    __syncthreads();

    for (uint32_t u = wave; u < nb; u += NW) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        uint32_t const q0 = a.starts[y], q1 = (ABL == 2) ? q0 : a.starts[y + 1], qi = (ABL == 2) ? q0 : q0 + nInside[y];
        R const* const A0 = (R const*)a.A + lc;
        R const* const X0 = (R const*)a.X + lc;
        struct Ops { R ar[4], ai[4], xr[4], xi[4]; };
        auto fetchA = [&](Ops& o, uint32_t q) {
            R const* Ab = A0 + size_t(ABL == 1 ? (q & 3) : a.pairs[2 * size_t(q)]) * 2 * LM * LM;
#pragma unroll
            for (int s = 0; s < 4; ++s) { int const k = 4 * s + lr; o.ar[s] = Ab[k * LM]; o.ai[s] = Ab[LM * LM + k * LM]; }
        };
        auto fetchXg = [&](Ops& o, uint32_t q) {
            R const* Xb = X0 + size_t(a.pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
            for (int s = 0; s < 4; ++s) { int const k = 4 * s + lr; o.xr[s] = Xb[k * LN]; o.xi[s] = Xb[P + k * LN]; }
        };
        auto readXl = [&](Ops& o, uint32_t q) {
            double const* Xb = lds + size_t(a.pairs[2 * size_t(q) + 1] - first) * 2 * P + lc;
#pragma unroll
            for (int s = 0; s < 4; ++s) { int const k = 4 * s + lr; o.xr[s] = Xb[k * LN]; o.xi[s] = Xb[P + k * LN]; }
        };
        auto mma = [&](Ops const& o) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                R const nai = -o.ai[s];
                cre = Acc<R>::mma(o.ar[s], o.xr[s], cre);
                cim = Acc<R>::mma(o.ar[s], o.xi[s], cim);
                cre = Acc<R>::mma(nai, o.xi[s], cre);
                cim = Acc<R>::mma(o.ai[s], o.xr[s], cim);
            }
        };
        // epilogue operands requested first (oldest in the queue: they have the whole product phase to arrive)
        EpiOps<R, EPI, 1, HASH> ops[4];
        auto row_of = [&](int r) { return Acc<R>::row(lane, r); };
#pragma unroll
        for (int r = 0; r < 4; ++r) ops[r].load(a, size_t(y) * 2 * P + row_of(r) * LN + lc, P);

        // pairs inside the chunk: A from global memory one pair ahead, X from LDS
        Ops o0, o1;
        uint32_t q = q0;
        if (q < qi) fetchA(o0, q);
        for (; q + 2 <= qi; q += 2) {
            fetchA(o1, q + 1);
            readXl(o0, q); mma(o0);
            if (q + 2 < qi) fetchA(o0, q + 2);
            readXl(o1, q + 1); mma(o1);
        }
        if (q < qi) { readXl(o0, q); mma(o0); ++q; }
        // the others: both operands from global memory
        if (q < q1) { fetchA(o0, q); fetchXg(o0, q); }
        for (; q + 2 <= q1; q += 2) {
            fetchA(o1, q + 1); fetchXg(o1, q + 1);
            mma(o0);
            if (q + 2 < q1) { fetchA(o0, q + 2); fetchXg(o0, q + 2); }
            mma(o1);
        }
        if (q < q1) mma(o0);

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX[y];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int const e = row_of(r) * LN + lc;
            size_t const off = size_t(y) * 2 * P + e;
            R yr[1] = {cre[r]}, yi[1] = {cim[r]};
            epilogue_row<R, EPI, 1, NPL, 1, HASH>(a, off, P, yr, yi, sr, si, 0, ops[r], bq, e, part, key);
        }
    }

    if constexpr (NPL > 0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            double v = part[p][0];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) red[(wave * 3 + p) * LN + lane] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += NW * 64) {
            int const p = e / LN, j = e % LN;
            double sum = 0;
            for (int w = 0; w < NW; ++w) sum += red[(w * 3 + p) * LN + j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// V2b: as V2, but the A operands form ONE stream over all pairs of all Y blocks of the wave, fetched exactly one pair
// ahead with no condition in the loop (the compiler can count the loads: no drain of the queue in front of the MFMAs);
// the X blocks that lie outside the chunk (at most NOUT per Y block in registers, more through a slow path) are requested
// at the start of the Y block, behind its epilogue operands.
template <int EPI, bool HASH, int CHMAX, int NOUT>
__global__ __launch_bounds__(256, 2) void k_spmm_ldsx_b(SpmmArgs a, uint32_t const* __restrict__ nInside) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    __shared__ double lds[CHMAX * 2 * P + 4 * 3 * LN];
    double* const red = lds + CHMAX * 2 * P;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;

    for (uint32_t b = wave; b < nb; b += 4) {
        double const* src = (double const*)a.X + size_t(first + b) * 2 * P;
#pragma unroll
        for (int piece = 0; piece < 4; ++piece)
            __builtin_amdgcn_global_load_lds(GPTR(src + piece * 128 + lane * 2), LPTR(lds + b * 2 * P + piece * 128), 16, 0, 0);
    }
    R sr[1], si[1];
    sr[0] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc];
    si[0] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1][1] = {};

    struct AOps { R ar[4], ai[4]; };
    struct XOps { R xr[4], xi[4]; };
    R const* const A0 = (R const*)a.A + lr * LM + lc;
    R const* const X0 = (R const*)a.X + lr * LN + lc;
    auto fetchA = [&](AOps& o, uint32_t q) {
        R const* Ab = A0 + size_t(a.pairs[2 * size_t(q)]) * 2 * LM * LM;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.ar[s] = Ab[4 * s * LM]; o.ai[s] = Ab[LM * LM + 4 * s * LM]; }
    };
    auto fetchXg = [&](XOps& o, uint32_t q) {
        R const* Xb = X0 + size_t(a.pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.xr[s] = Xb[4 * s * LN]; o.xi[s] = Xb[P + 4 * s * LN]; }
    };
    // the first pair of the wave's stream (pairs of a Y block are never empty here? they may be: skip empty blocks)
    uint32_t uNext = wave;
    auto first_pair_from = [&](uint32_t u) -> uint32_t {   // first unit >= u (step 4) that has pairs; nb if none
        while (u < nb && a.starts[first + u] == a.starts[first + u + 1]) u += 4;
        return u;
    };
    AOps nxt;
    {
        uint32_t const u0 = first_pair_from(wave);
        if (u0 < nb) fetchA(nxt, a.starts[first + u0]);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the DMAs (and the first A operands) have landed
    __syncthreads();

    for (uint32_t u = wave; u < nb; u += 4) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        uint32_t const q0 = a.starts[y], q1 = a.starts[y + 1], qi = q0 + nInside[y];
        EpiOps<R, EPI, 1, HASH> ops[4];
        auto row_of = [&](int r) { return Acc<R>::row(lane, r); };
#pragma unroll
        for (int r = 0; r < 4; ++r) ops[r].load(a, size_t(y) * 2 * P + row_of(r) * LN + lc, P);
        XOps xo[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) if (qi + j < q1) fetchXg(xo[j], qi + j);
        // where the A stream continues behind this Y block
        uint32_t const un = first_pair_from(u + 4);
        uint32_t const qNextUnit = (un < nb) ? a.starts[first + un] : (q1 > q0 ? q1 - 1 : 0);   // clamp: one redundant load at the very end
        for (uint32_t q = q0; q < q1; ++q) {
            AOps const cur = nxt;
            fetchA(nxt, (q + 1 < q1) ? q + 1 : qNextUnit);
            XOps x;
            if (q < qi) {
                double const* Xb = lds + size_t(a.pairs[2 * size_t(q) + 1] - first) * 2 * P + lr * LN + lc;
#pragma unroll
                for (int s = 0; s < 4; ++s) { x.xr[s] = Xb[4 * s * LN]; x.xi[s] = Xb[P + 4 * s * LN]; }
            } else {
                uint32_t const j = q - qi;
                bool done = false;
#pragma unroll
                for (int jj = 0; jj < NOUT; ++jj) if (j == uint32_t(jj)) { x = xo[jj]; done = true; }
                if (!done) fetchXg(x, q);      // rare: more than NOUT blocks from outside the chunk
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                R const nai = -cur.ai[s];
                cre = Acc<R>::mma(cur.ar[s], x.xr[s], cre);
                cim = Acc<R>::mma(cur.ar[s], x.xi[s], cim);
                cre = Acc<R>::mma(nai, x.xi[s], cre);
                cim = Acc<R>::mma(cur.ai[s], x.xr[s], cim);
            }
        }
        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX[y];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int const e = row_of(r) * LN + lc;
            size_t const off = size_t(y) * 2 * P + e;
            R yr[1] = {cre[r]}, yi[1] = {cim[r]};
            epilogue_row<R, EPI, 1, NPL, 1, HASH>(a, off, P, yr, yi, sr, si, 0, ops[r], bq, e, part, key);
        }
    }

    if constexpr (NPL > 0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            double v = part[p][0];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) red[(wave * 3 + p) * LN + lane] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, j = e % LN;
            double const sum = ((red[(0 * 3 + p) * LN + j] + red[(1 * 3 + p) * LN + j]) + red[(2 * 3 + p) * LN + j]) + red[(3 * 3 + p) * LN + j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// V3: X patch in LDS as V2; the A operands are one stream over all pairs of the wave, always ONE pair ahead, fetched without
// any condition (clamped at the very end) into two named register sets P/Q, so that every wait the compiler inserts is an
// exact count; an odd pair count is repaired by one register copy at the end of the Y block.  At most NOUT X blocks from
// outside the chunk sit in registers per Y block (requested at its start), further ones go through a slow path.
template <int EPI, bool HASH, int CHMAX, int NOUT, int NW, int WPS = (2 * NW) / 4>
__global__ __launch_bounds__(NW * 64, WPS) void k_spmm_v3(SpmmArgs a, uint32_t const* __restrict__ nInside) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    __shared__ double lds[CHMAX * 2 * P + NW * 3 * LN];
    double* const red = lds + CHMAX * 2 * P;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    // index lists through the constant address space: uniform reads of them become scalar loads (s_load, lgkmcnt) even with
    // stores to other arrays in the loop -- as vector loads they would sit in the vmcnt queue in front of the operand stream
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts; CU32 const nIn = (CU32)(uintptr_t)nInside;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;

    for (uint32_t b = wave; b < nb; b += NW) {
        double const* src = (double const*)a.X + size_t(first + b) * 2 * P;
#pragma unroll
        for (int piece = 0; piece < 4; ++piece)
            glds16(src + piece * 128 + lane * 2, lds_addr(lds + b * 2 * P + piece * 128));
    }
    R sr[1], si[1];
    sr[0] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc];
    si[0] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1][1] = {};

    struct AOps { R ar[4], ai[4]; };
    struct XOps { R xr[4], xi[4]; };
    R const* const A0 = (R const*)a.A + lr * LM + lc;
    R const* const X0 = (R const*)a.X + lr * LN + lc;
    double const* const L0 = lds + lr * LN + lc;
    auto fetchA = [&](AOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * LM * LM;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.ar[s] = Ab[4 * s * LM]; o.ai[s] = Ab[LM * LM + 4 * s * LM]; }
    };
    auto fetchXg = [&](XOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.xr[s] = Xb[4 * s * LN]; o.xi[s] = Xb[P + 4 * s * LN]; }
    };
    auto next_unit = [&](uint32_t u) -> uint32_t {   // first unit >= u (step NW) that has pairs; nb if none
        while (u < nb && starts[first + u] == starts[first + u + 1]) u += NW;
        return u;
    };
    AOps Pa, Qa;
    uint32_t lastValid = 0;
    {
        uint32_t const u0 = next_unit(wave);
        lastValid = (u0 < nb) ? starts[first + u0] : 0;
        fetchA(Pa, lastValid);             // harmless if the wave has no pairs at all: pair 0 exists whenever nPairs > 0
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMAs have landed (the compiler does not know about them)
    __syncthreads();

    for (uint32_t u = wave; u < nb; u += NW) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], q1 = starts[y + 1], qi = q0 + nIn[y];
        EpiOps<R, EPI, 1, HASH> ops[4];
        auto row_of = [&](int r) { return Acc<R>::row(lane, r); };
#pragma unroll
        for (int r = 0; r < 4; ++r) ops[r].load(a, size_t(y) * 2 * P + row_of(r) * LN + lc, P);
        XOps xo[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) if (qi + j < q1) fetchXg(xo[j], qi + j);
        uint32_t const qFast = (q1 - qi > NOUT) ? qi + NOUT : q1;          // pairs [qFast, q1) take the slow path
        uint32_t const un = next_unit(u + NW);
        uint32_t const qNext = (un < nb) ? starts[first + un] : lastValid;   // where the A stream continues (clamped at the end)
        if (q1 > q0) lastValid = q1 - 1;

        auto xof = [&](XOps& x, uint32_t q) __attribute__((always_inline)) {
            if (q < qi) {
                double const* Xb = L0 + size_t(pairs[2 * size_t(q) + 1] - first) * 2 * P;
#pragma unroll
                for (int s = 0; s < 4; ++s) { x.xr[s] = Xb[4 * s * LN]; x.xi[s] = Xb[P + 4 * s * LN]; }
            } else {
                uint32_t const j = q - qi;
#pragma unroll
                for (int jj = 0; jj < NOUT; ++jj) if (j == uint32_t(jj)) x = xo[jj];
            }
        };
        auto mma = [&](AOps const& A, XOps const& x) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                R const nai = -A.ai[s];
                cre = Acc<R>::mma(A.ar[s], x.xr[s], cre);
                cim = Acc<R>::mma(A.ar[s], x.xi[s], cim);
                cre = Acc<R>::mma(nai, x.xi[s], cre);
                cim = Acc<R>::mma(A.ai[s], x.xr[s], cim);
            }
        };
        uint32_t q = q0;
        // the A stream covers the pairs [q0, qFast) of every Y block; Pa holds A(q0) (in flight or landed)
        for (; q + 2 <= qFast; q += 2) {
            fetchA(Qa, q + 1);
            { XOps x; xof(x, q); mma(Pa, x); }
            __builtin_amdgcn_sched_barrier(0);
            fetchA(Pa, (q + 2 < qFast) ? q + 2 : qNext);
            { XOps x; xof(x, q + 1); mma(Qa, x); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < qFast) {
            fetchA(Qa, qNext);
            { XOps x; xof(x, q); mma(Pa, x); }
            __builtin_amdgcn_sched_barrier(0);
            Pa = Qa;
            ++q;
        } else if (q0 == qFast) {
            // no fast pairs in this block: Pa already holds A(qNext') of an earlier request -- it was requested for THIS q0;
            // re-request for the next unit (rare: blocks with only slow pairs do not exist, qFast > q0 whenever q1 > q0)
        }
        for (; q < q1; ++q) {                 // more than NOUT blocks from outside the chunk (rare)
            AOps As; XOps x;
            fetchA(As, q); fetchXg(x, q);
            mma(As, x);
        }
        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX[y];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int const e = row_of(r) * LN + lc;
            size_t const off = size_t(y) * 2 * P + e;
            R yr[1] = {cre[r]}, yi[1] = {cim[r]};
            epilogue_row<R, EPI, 1, NPL, 1, HASH>(a, off, P, yr, yi, sr, si, 0, ops[r], bq, e, part, key);
        }
    }

    if constexpr (NPL > 0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            double v = part[p][0];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) red[(wave * 3 + p) * LN + lane] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += NW * 64) {
            int const p = e / LN, j = e % LN;
            double sum = 0;
            for (int w = 0; w < NW; ++w) sum += red[(w * 3 + p) * LN + j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
    }
}

// V4 = V3 + the epilogue operands of the NEXT Y block are requested at the start of the current one (two blocks of
// HBM loads in flight per wave).  V3: X patch in LDS as V2; the A operands are one stream over all pairs of the wave, always ONE pair ahead, fetched without
// any condition (clamped at the very end) into two named register sets P/Q, so that every wait the compiler inserts is an
// exact count; an odd pair count is repaired by one register copy at the end of the Y block.  At most NOUT X blocks from
// outside the chunk sit in registers per Y block (requested at its start), further ones go through a slow path.
template <int EPI, bool HASH, int CHMAX, int NOUT, int NW, int WPS = (2 * NW) / 4>
__global__ __launch_bounds__(NW * 64, WPS) void k_spmm_v4(SpmmArgs a, uint32_t const* __restrict__ nInside) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    __shared__ double lds[CHMAX * 2 * P + NW * 3 * LN];
    double* const red = lds + CHMAX * 2 * P;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    // index lists through the constant address space: uniform reads of them become scalar loads (s_load, lgkmcnt) even with
    // stores to other arrays in the loop -- as vector loads they would sit in the vmcnt queue in front of the operand stream
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts; CU32 const nIn = (CU32)(uintptr_t)nInside;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;

    for (uint32_t b = wave; b < nb; b += NW) {
        double const* src = (double const*)a.X + size_t(first + b) * 2 * P;
#pragma unroll
        for (int piece = 0; piece < 4; ++piece)
            glds16(src + piece * 128 + lane * 2, lds_addr(lds + b * 2 * P + piece * 128));
    }
    R sr[1], si[1];
    sr[0] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc];
    si[0] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1][1] = {};

    struct AOps { R ar[4], ai[4]; };
    struct XOps { R xr[4], xi[4]; };
    R const* const A0 = (R const*)a.A + lr * LM + lc;
    R const* const X0 = (R const*)a.X + lr * LN + lc;
    double const* const L0 = lds + lr * LN + lc;
    auto fetchA = [&](AOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * LM * LM;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.ar[s] = Ab[4 * s * LM]; o.ai[s] = Ab[LM * LM + 4 * s * LM]; }
    };
    auto fetchXg = [&](XOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.xr[s] = Xb[4 * s * LN]; o.xi[s] = Xb[P + 4 * s * LN]; }
    };
    auto next_unit = [&](uint32_t u) -> uint32_t {   // first unit >= u (step NW) that has pairs; nb if none
        while (u < nb && starts[first + u] == starts[first + u + 1]) u += NW;
        return u;
    };
    AOps Pa, Qa;
    uint32_t lastValid = 0;
    {
        uint32_t const u0 = next_unit(wave);
        lastValid = (u0 < nb) ? starts[first + u0] : 0;
        fetchA(Pa, lastValid);             // harmless if the wave has no pairs at all: pair 0 exists whenever nPairs > 0
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMAs have landed (the compiler does not know about them)
    __syncthreads();

    EpiOps<R, EPI, 1, HASH> opsNext[4];
    if (wave < nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) opsNext[r].load(a, size_t(first + wave) * 2 * P + Acc<R>::row(lane, r) * LN + lc, P);
    }
    for (uint32_t u = wave; u < nb; u += NW) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], q1 = starts[y + 1], qi = q0 + nIn[y];
        auto row_of = [&](int r) { return Acc<R>::row(lane, r); };
        EpiOps<R, EPI, 1, HASH> ops[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ops[r] = opsNext[r];
        {   // the operands of the wave's next Y block (clamped to this one at the end: a re-read from the cache)
            uint32_t const yn = (u + NW < nb) ? y + NW : y;
#pragma unroll
            for (int r = 0; r < 4; ++r) opsNext[r].load(a, size_t(yn) * 2 * P + row_of(r) * LN + lc, P);
        }
        XOps xo[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) if (qi + j < q1) fetchXg(xo[j], qi + j);
        uint32_t const qFast = (q1 - qi > NOUT) ? qi + NOUT : q1;          // pairs [qFast, q1) take the slow path
        uint32_t const un = next_unit(u + NW);
        uint32_t const qNext = (un < nb) ? starts[first + un] : lastValid;   // where the A stream continues (clamped at the end)
        if (q1 > q0) lastValid = q1 - 1;

        auto xof = [&](XOps& x, uint32_t q) __attribute__((always_inline)) {
            if (q < qi) {
                double const* Xb = L0 + size_t(pairs[2 * size_t(q) + 1] - first) * 2 * P;
#pragma unroll
                for (int s = 0; s < 4; ++s) { x.xr[s] = Xb[4 * s * LN]; x.xi[s] = Xb[P + 4 * s * LN]; }
            } else {
                uint32_t const j = q - qi;
#pragma unroll
                for (int jj = 0; jj < NOUT; ++jj) if (j == uint32_t(jj)) x = xo[jj];
            }
        };
        auto mma = [&](AOps const& A, XOps const& x) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                R const nai = -A.ai[s];
                cre = Acc<R>::mma(A.ar[s], x.xr[s], cre);
                cim = Acc<R>::mma(A.ar[s], x.xi[s], cim);
                cre = Acc<R>::mma(nai, x.xi[s], cre);
                cim = Acc<R>::mma(A.ai[s], x.xr[s], cim);
            }
        };
        uint32_t q = q0;
        // the A stream covers the pairs [q0, qFast) of every Y block; Pa holds A(q0) (in flight or landed)
        for (; q + 2 <= qFast; q += 2) {
            fetchA(Qa, q + 1);
            { XOps x; xof(x, q); mma(Pa, x); }
            __builtin_amdgcn_sched_barrier(0);
            fetchA(Pa, (q + 2 < qFast) ? q + 2 : qNext);
            { XOps x; xof(x, q + 1); mma(Qa, x); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < qFast) {
            fetchA(Qa, qNext);
            { XOps x; xof(x, q); mma(Pa, x); }
            __builtin_amdgcn_sched_barrier(0);
            Pa = Qa;
            ++q;
        } else if (q0 == qFast) {
            // no fast pairs in this block: Pa already holds A(qNext') of an earlier request -- it was requested for THIS q0;
            // re-request for the next unit (rare: blocks with only slow pairs do not exist, qFast > q0 whenever q1 > q0)
        }
        for (; q < q1; ++q) {                 // more than NOUT blocks from outside the chunk (rare)
            AOps As; XOps x;
            fetchA(As, q); fetchXg(x, q);
            mma(As, x);
        }
        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX[y];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int const e = row_of(r) * LN + lc;
            size_t const off = size_t(y) * 2 * P + e;
            R yr[1] = {cre[r]}, yi[1] = {cim[r]};
            epilogue_row<R, EPI, 1, NPL, 1, HASH>(a, off, P, yr, yi, sr, si, 0, ops[r], bq, e, part, key);
        }
    }

    if constexpr (NPL > 0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            double v = part[p][0];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) red[(wave * 3 + p) * LN + lane] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += NW * 64) {
            int const p = e / LN, j = e % LN;
            double sum = 0;
            for (int w = 0; w < NW; ++w) sum += red[(w * 3 + p) * LN + j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// V5: phase-split.  vmcnt retires in order, so a wave that mixes HBM-latency loads (epilogue operands, X blocks from outside
// the chunk) with the short-latency A stream stalls on the slowest one at every Y block.  Here a wave owns up to UMAX Y blocks
// of the chunk and keeps ALL their accumulators in registers:
//   phase 1: the products whose X block is in the LDS patch, for all its Y blocks -- only A loads in the queue;
//   phase 2: the products whose X block lies outside the chunk -- A and X from global memory, one pair ahead;
//   phase 3: the epilogues, a pure stream: operands of the next Y block requested before the current one is finished.
// The two waves of a SIMD run their phases independently, one streams while the other multiplies.
template <int EPI, bool HASH, int CHMAX, int NW, int UG, int WPS = (2 * NW) / 4>
__global__ __launch_bounds__(NW * 64, WPS) void k_spmm_v5(SpmmArgs a, uint32_t const* __restrict__ nInside) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N, UMAX = UG, NG = CHMAX / (NW * UG);
    using T4 = d4;
    __shared__ double lds[CHMAX * 2 * P + NW * 3 * LN];
    double* const red = lds + CHMAX * 2 * P;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts; CU32 const nIn = (CU32)(uintptr_t)nInside;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;

    for (uint32_t b = wave; b < nb; b += NW) {
        double const* src = (double const*)a.X + size_t(first + b) * 2 * P;
#pragma unroll
        for (int piece = 0; piece < 4; ++piece) glds16(src + piece * 128 + lane * 2, lds_addr(lds + b * 2 * P + piece * 128));
    }
    R sr[1], si[1];
    sr[0] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc];
    si[0] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1][1] = {};

    struct AOps { R ar[4], ai[4]; };
    struct XOps { R xr[4], xi[4]; };
    R const* const A0 = (R const*)a.A + lr * LM + lc;
    R const* const X0 = (R const*)a.X + lr * LN + lc;
    double const* const L0 = lds + lr * LN + lc;
    auto fetchA = [&](AOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * LM * LM;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.ar[s] = Ab[4 * s * LM]; o.ai[s] = Ab[LM * LM + 4 * s * LM]; }
    };
    auto fetchXg = [&](XOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int s = 0; s < 4; ++s) { o.xr[s] = Xb[4 * s * LN]; o.xi[s] = Xb[P + 4 * s * LN]; }
    };
    auto readXl = [&](XOps& x, uint32_t q) __attribute__((always_inline)) {
        double const* Xb = L0 + size_t(pairs[2 * size_t(q) + 1] - first) * 2 * P;
#pragma unroll
        for (int s = 0; s < 4; ++s) { x.xr[s] = Xb[4 * s * LN]; x.xi[s] = Xb[P + 4 * s * LN]; }
    };
#pragma unroll
    for (int g = 0; g < NG; ++g) {
    // the pair ranges of this wave's Y blocks (scalars): [q0, qi) in the patch, [qi, q1) outside
    uint32_t q0[UMAX], qi[UMAX], q1[UMAX];
#pragma unroll
    for (int i = 0; i < UMAX; ++i) {
        uint32_t const u = wave + (g * UG + i) * NW;
        if (u < nb) { q0[i] = starts[first + u]; q1[i] = starts[first + u + 1]; qi[i] = q0[i] + nIn[first + u]; }
        else { q0[i] = qi[i] = q1[i] = 0; }
    }
    T4 cre[UMAX], cim[UMAX];
#pragma unroll
    for (int i = 0; i < UMAX; ++i) { cre[i] = T4{0, 0, 0, 0}; cim[i] = T4{0, 0, 0, 0}; }
    auto mma = [&](T4& re, T4& im, AOps const& A, XOps const& x) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            R const nai = -A.ai[s];
            re = Acc<R>::mma(A.ar[s], x.xr[s], re);
            im = Acc<R>::mma(A.ar[s], x.xi[s], im);
            re = Acc<R>::mma(nai, x.xi[s], re);
            im = Acc<R>::mma(A.ai[s], x.xr[s], im);
        }
    };
    uint32_t anyPair = 0;   // some valid pair index for the clamped request at the end of a stream
#pragma unroll
    for (int i = 0; i < UMAX; ++i) if (q1[i] > q0[i]) anyPair = q0[i];

    // ---- phase 1: products with X in the patch ---------------------------------------------------------------------
    {
        AOps Pa, Qa;
        uint32_t firstQ = anyPair;
#pragma unroll
        for (int i = UMAX - 1; i >= 0; --i) if (qi[i] > q0[i]) firstQ = q0[i];
        fetchA(Pa, firstQ);
        if (g == 0) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the DMAs of the patch (older than the 8 loads above) have landed
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < UMAX; ++i) {
            uint32_t qNext = anyPair;                        // first pair of the following Y blocks' patch products (clamped)
#pragma unroll
            for (int j = UMAX - 1; j > i; --j) if (qi[j] > q0[j]) qNext = q0[j];
            uint32_t q = q0[i];
            for (; q + 2 <= qi[i]; q += 2) {
                fetchA(Qa, q + 1);
                { XOps x; readXl(x, q); mma(cre[i], cim[i], Pa, x); }
                __builtin_amdgcn_sched_barrier(0);
                fetchA(Pa, (q + 2 < qi[i]) ? q + 2 : qNext);
                { XOps x; readXl(x, q + 1); mma(cre[i], cim[i], Qa, x); }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q < qi[i]) {
                fetchA(Qa, qNext);
                { XOps x; readXl(x, q); mma(cre[i], cim[i], Pa, x); }
                __builtin_amdgcn_sched_barrier(0);
                Pa = Qa;
            }
        }
    }
    // ---- phase 2: products with X outside the patch ----------------------------------------------------------------
    {
        AOps Pa, Qa; XOps Px, Qx;
        uint32_t firstQ = anyPair;
#pragma unroll
        for (int i = UMAX - 1; i >= 0; --i) if (q1[i] > qi[i]) firstQ = qi[i];
        fetchA(Pa, firstQ); fetchXg(Px, firstQ);
#pragma unroll
        for (int i = 0; i < UMAX; ++i) {
            uint32_t qNext = anyPair;
#pragma unroll
            for (int j = UMAX - 1; j > i; --j) if (q1[j] > qi[j]) qNext = qi[j];
            uint32_t q = qi[i];
            for (; q + 2 <= q1[i]; q += 2) {
                fetchA(Qa, q + 1); fetchXg(Qx, q + 1);
                mma(cre[i], cim[i], Pa, Px);
                __builtin_amdgcn_sched_barrier(0);
                uint32_t const qn = (q + 2 < q1[i]) ? q + 2 : qNext;
                fetchA(Pa, qn); fetchXg(Px, qn);
                mma(cre[i], cim[i], Qa, Qx);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (q < q1[i]) {
                fetchA(Qa, qNext); fetchXg(Qx, qNext);
                mma(cre[i], cim[i], Pa, Px);
                __builtin_amdgcn_sched_barrier(0);
                Pa = Qa; Px = Qx;
            }
        }
    }
    // ---- phase 3: epilogues, streamed ------------------------------------------------------------------------------
    {
        auto row_of = [&](int r) { return Acc<R>::row(lane, r); };
        EpiOps<R, EPI, 1, HASH> ops[4];
        auto request = [&](int i) __attribute__((always_inline)) {
            uint32_t const u = wave + (g * UG + i) * NW;
            uint32_t const y = first + ((u < nb) ? u : wave);      // clamped: a re-read
#pragma unroll
            for (int r = 0; r < 4; ++r) ops[r].load(a, size_t(y) * 2 * P + row_of(r) * LN + lc, P);
        };
        request(0);
#pragma unroll
        for (int i = 0; i < UMAX; ++i) {
            uint32_t const u = wave + (g * UG + i) * NW;
            EpiOps<R, EPI, 1, HASH> cur[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cur[r] = ops[r];
            if (i + 1 < UMAX) request(i + 1);            // in flight while this block is finished and stored
            if (u < nb) {
                uint32_t const y = first + u;
                uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
                uint32_t bq = 0xffffffffu;
                if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX[y];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int const e = row_of(r) * LN + lc;
                    size_t const off = size_t(y) * 2 * P + e;
                    R yr[1] = {cre[i][r]}, yi[1] = {cim[i][r]};
                    epilogue_row<R, EPI, 1, NPL, 1, HASH>(a, off, P, yr, yi, sr, si, 0, cur[r], bq, e, part, key);
                }
            }
        }
    }

    }
    if constexpr (NPL > 0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            double v = part[p][0];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 16) red[(wave * 3 + p) * LN + lane] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += NW * 64) {
            int const p = e / LN, j = e % LN;
            double sum = 0;
            for (int w = 0; w < NW; ++w) sum += red[(w * 3 + p) * LN + j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// V6: the shipped structure (no LDS, one 16 x 16 strip per wave, two register sets) on a ROW-PAIR-INTERLEAVED block layout:
// plane[i/2][j][i%2] for X-shaped blocks, plane[k/2][p(i)][k%2] for A blocks (p = the row permutation below), so that every
// operand load, every epilogue load and every store moves 16 bytes per lane (1 KiB per wave instruction instead of 512 B).
//   k-steps: lane group lr loads the k pairs lr and lr + 4 -> k = 2 lr, 2 lr + 1, 2 lr + 8, 2 lr + 9 for MFMA steps 0..3
//   rows:    the A operand of lane column a is row ROWP(a) = 2 (a % 4 + 4 (a / 8)) + (a / 4) % 2, so that accumulator
//            registers (0, 1) and (2, 3) of lane group lr are the row pairs (2 lr, 2 lr + 1) and (2 lr + 8, 2 lr + 9)
__device__ inline int rowp(int a) { return 2 * ((a & 3) + 4 * (a >> 3)) + ((a >> 2) & 1); }
using d2v = __attribute__((ext_vector_type(2))) double;

template <int EPI, bool HASH>
__global__ __launch_bounds__(256, 2) void k_spmm_v6(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    R const sr = (EPI == EPI_NONE) ? 0 : ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc], si = (EPI == EPI_NONE) ? 0 : ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1] = {};
    __shared__ double s[4][3][LN];

    struct Ops { d2v ar[2], ai[2], xr[2], xi[2]; };   // [k-pair lr | lr + 4]
    // element offsets of this lane: A (k-pair kp, row ROWP(lc)), X (k-pair kp, column lc)
    R const* const A0 = (R const*)a.A + (lr * 16 + rowp(lc)) * 2;
    R const* const X0 = (R const*)a.X + (lr * 16 + lc) * 2;
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * P;
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            o.ar[h] = *(d2v const*)(Ab + h * 128); o.ai[h] = *(d2v const*)(Ab + P + h * 128);
            o.xr[h] = *(d2v const*)(Xb + h * 128); o.xi[h] = *(d2v const*)(Xb + P + h * 128);
        }
    };
    for (uint32_t u = wave; u < last - first; u += 4) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    R const nai = -o.ai[h][e];
                    cre = Acc<R>::mma(o.ar[h][e], o.xr[h][e], cre);
                    cim = Acc<R>::mma(o.ar[h][e], o.xi[h][e], cim);
                    cre = Acc<R>::mma(nai, o.xi[h][e], cre);
                    cim = Acc<R>::mma(o.ai[h][e], o.xr[h][e], cim);
                }
        };
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        Ops o0, o1;
        if (q0 < q1) fetch(o0, q0);
        if (q0 + 1 < q1) fetch(o1, q0 + 1);
        // epilogue operands: row pairs (2 lr, 2 lr + 1) and (2 lr + 8, 2 lr + 9) of column lc, 16 bytes each
        size_t const eoff[2] = { size_t(y) * 2 * P + (lr * 16 + lc) * 2, size_t(y) * 2 * P + ((lr + 4) * 16 + lc) * 2 };
        d2v ur[2], ui[2], xr[2], xi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) if constexpr (EPI != EPI_NONE) {
            ur[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h])); ui[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h] + P));
            if constexpr (EPI == EPI_XPAY_DOT) { xr[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h])); xi[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h] + P)); }
        }
        uint32_t q = q0;
        for (; q + 2 <= q1; q += 2) {
            mma(o0);
            if (q + 2 < q1) fetch(o0, q + 2);
            mma(o1);
            if (q + 3 < q1) fetch(o1, q + 3);
        }
        if (q < q1) mma(o0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2v yr, yi, nr, ni;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                yr[e] = cre[2 * h + e]; yi[e] = cim[2 * h + e];
                if constexpr (EPI == EPI_NONE) continue;
                if constexpr (EPI == EPI_XPAY_DOT) {
                    R const tr = xr[h][e] + sr * ur[h][e] - si * ui[h][e], ti = xi[h][e] + si * ur[h][e] + sr * ui[h][e];
                    nr[e] = yr[e] + sr * tr - si * ti; ni[e] = yi[e] + si * tr + sr * ti;
                } else {
                    nr[e] = sr * yr[e] - si * yi[e] + ur[h][e]; ni[e] = si * yr[e] + sr * yi[e] + ui[h][e];
                }
                int const row = 2 * (lr + 4 * h) + e;                    // logical element (row, lc) for the shadow vector
                double const wr = shadow_value(key, uint32_t(row * LN + lc)), wi = shadow_value(key, uint32_t(P + row * LN + lc));
                double const dr = nr[e], di = ni[e];
                part[0] += dr * wr - di * wi;
                part[1] += dr * wi + di * wr;
                if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] += dr * dr + di * di;
            }
            __builtin_nontemporal_store(yr, (d2v*)((R*)a.Y + eoff[h])); __builtin_nontemporal_store(yi, (d2v*)((R*)a.Y + eoff[h] + P));
            if constexpr (EPI != EPI_NONE) { __builtin_nontemporal_store(nr, (d2v*)((R*)a.e0 + eoff[h])); __builtin_nontemporal_store(ni, (d2v*)((R*)a.e0 + eoff[h] + P)); }
        }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        double v = part[p];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) s[wave][p][lane] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NPL * LN; e += 256) {
        int const p = e / LN, j = e % LN;
        double const sum = ((s[0][p][j] + s[1][p][j]) + s[2][p][j]) + s[3][p][j];
        write_record<EPI>(a, chunk, LN, p, j, sum);
    }
}

template <int EPI, bool HASH, int ANT>
__global__ __launch_bounds__(256, 2) void k_spmm_v6nt(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    R const sr = (EPI == EPI_NONE) ? 0 : ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc], si = (EPI == EPI_NONE) ? 0 : ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1] = {};
    __shared__ double s[4][3][LN];

    struct Ops { d2v ar[2], ai[2], xr[2], xi[2]; };   // [k-pair lr | lr + 4]
    // element offsets of this lane: A (k-pair kp, row ROWP(lc)), X (k-pair kp, column lc)
    R const* const A0 = (R const*)a.A + (lr * 16 + rowp(lc)) * 2;
    R const* const X0 = (R const*)a.X + (lr * 16 + lc) * 2;
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * P;
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            o.ar[h] = __builtin_nontemporal_load((d2v const*)(Ab + h * 128)); o.ai[h] = __builtin_nontemporal_load((d2v const*)(Ab + P + h * 128));
            if (ANT > 1) { o.xr[h] = __builtin_nontemporal_load((d2v const*)(Xb + h * 128)); o.xi[h] = __builtin_nontemporal_load((d2v const*)(Xb + P + h * 128)); }
            else { o.xr[h] = *(d2v const*)(Xb + h * 128); o.xi[h] = *(d2v const*)(Xb + P + h * 128); }
        }
    };
    for (uint32_t u = wave; u < last - first; u += 4) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    R const nai = -o.ai[h][e];
                    cre = Acc<R>::mma(o.ar[h][e], o.xr[h][e], cre);
                    cim = Acc<R>::mma(o.ar[h][e], o.xi[h][e], cim);
                    cre = Acc<R>::mma(nai, o.xi[h][e], cre);
                    cim = Acc<R>::mma(o.ai[h][e], o.xr[h][e], cim);
                }
        };
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        Ops o0, o1;
        if (q0 < q1) fetch(o0, q0);
        if (q0 + 1 < q1) fetch(o1, q0 + 1);
        // epilogue operands: row pairs (2 lr, 2 lr + 1) and (2 lr + 8, 2 lr + 9) of column lc, 16 bytes each
        size_t const eoff[2] = { size_t(y) * 2 * P + (lr * 16 + lc) * 2, size_t(y) * 2 * P + ((lr + 4) * 16 + lc) * 2 };
        d2v ur[2], ui[2], xr[2], xi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) if constexpr (EPI != EPI_NONE) {
            ur[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h])); ui[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h] + P));
            if constexpr (EPI == EPI_XPAY_DOT) { xr[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h])); xi[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h] + P)); }
        }
        uint32_t q = q0;
        for (; q + 2 <= q1; q += 2) {
            mma(o0);
            if (q + 2 < q1) fetch(o0, q + 2);
            mma(o1);
            if (q + 3 < q1) fetch(o1, q + 3);
        }
        if (q < q1) mma(o0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2v yr, yi, nr, ni;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                yr[e] = cre[2 * h + e]; yi[e] = cim[2 * h + e];
                if constexpr (EPI == EPI_NONE) continue;
                if constexpr (EPI == EPI_XPAY_DOT) {
                    R const tr = xr[h][e] + sr * ur[h][e] - si * ui[h][e], ti = xi[h][e] + si * ur[h][e] + sr * ui[h][e];
                    nr[e] = yr[e] + sr * tr - si * ti; ni[e] = yi[e] + si * tr + sr * ti;
                } else {
                    nr[e] = sr * yr[e] - si * yi[e] + ur[h][e]; ni[e] = si * yr[e] + sr * yi[e] + ui[h][e];
                }
                int const row = 2 * (lr + 4 * h) + e;                    // logical element (row, lc) for the shadow vector
                double const wr = shadow_value(key, uint32_t(row * LN + lc)), wi = shadow_value(key, uint32_t(P + row * LN + lc));
                double const dr = nr[e], di = ni[e];
                part[0] += dr * wr - di * wi;
                part[1] += dr * wi + di * wr;
                if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] += dr * dr + di * di;
            }
            __builtin_nontemporal_store(yr, (d2v*)((R*)a.Y + eoff[h])); __builtin_nontemporal_store(yi, (d2v*)((R*)a.Y + eoff[h] + P));
            if constexpr (EPI != EPI_NONE) { __builtin_nontemporal_store(nr, (d2v*)((R*)a.e0 + eoff[h])); __builtin_nontemporal_store(ni, (d2v*)((R*)a.e0 + eoff[h] + P)); }
        }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        double v = part[p];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) s[wave][p][lane] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NPL * LN; e += 256) {
        int const p = e / LN, j = e % LN;
        double const sum = ((s[0][p][j] + s[1][p][j]) + s[2][p][j]) + s[3][p][j];
        write_record<EPI>(a, chunk, LN, p, j, sum);
    }
}

// ---------------------------------------------------------------------------------------------------
// V7 = V3 (X patch in LDS, A stream one pair ahead with exact waits, at most NOUT outside X blocks in registers) on the
// row-pair-interleaved layout of V6: every access is 16 bytes per lane, LDS reads are ds_read_b128.
template <int EPI, bool HASH, int CHMAX, int NOUT, int NW, int WPS = (2 * NW) / 4>
__global__ __launch_bounds__(NW * 64, WPS) void k_spmm_v7(SpmmArgs a, uint32_t const* __restrict__ nInside) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    __shared__ double lds[CHMAX * 2 * P + NW * 3 * LN];
    double* const red = lds + CHMAX * 2 * P;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts; CU32 const nIn = (CU32)(uintptr_t)nInside;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;
    for (uint32_t b = wave; b < nb; b += NW) {
        double const* src = (double const*)a.X + size_t(first + b) * 2 * P;
#pragma unroll
        for (int piece = 0; piece < 4; ++piece) glds16(src + piece * 128 + lane * 2, lds_addr(lds + b * 2 * P + piece * 128));
    }
    R const sr = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc], si = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1] = {};

    struct AOps { d2v ar[2], ai[2]; };
    struct XOps { d2v xr[2], xi[2]; };
    R const* const A0 = (R const*)a.A + (lr * 16 + rowp(lc)) * 2;
    R const* const X0 = (R const*)a.X + (lr * 16 + lc) * 2;
    double const* const L0 = lds + (lr * 16 + lc) * 2;
    auto fetchA = [&](AOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * P;
#pragma unroll
        for (int h = 0; h < 2; ++h) { o.ar[h] = *(d2v const*)(Ab + h * 128); o.ai[h] = *(d2v const*)(Ab + P + h * 128); }
    };
    auto fetchXg = [&](XOps& o, uint32_t q) __attribute__((always_inline)) {
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int h = 0; h < 2; ++h) { o.xr[h] = *(d2v const*)(Xb + h * 128); o.xi[h] = *(d2v const*)(Xb + P + h * 128); }
    };
    auto next_unit = [&](uint32_t u) -> uint32_t {
        while (u < nb && starts[first + u] == starts[first + u + 1]) u += NW;
        return u;
    };
    AOps Pa, Qa;
    uint32_t lastValid = 0;
    {
        uint32_t const u0 = next_unit(wave);
        lastValid = (u0 < nb) ? starts[first + u0] : 0;
        fetchA(Pa, lastValid);
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // the DMAs (older than the 4 loads above) have landed
    __syncthreads();

    for (uint32_t u = wave; u < nb; u += NW) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], q1 = starts[y + 1], qi = q0 + nIn[y];
        size_t const eoff[2] = { size_t(y) * 2 * P + (lr * 16 + lc) * 2, size_t(y) * 2 * P + ((lr + 4) * 16 + lc) * 2 };
        d2v ur[2], ui[2], vr[2], vi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ur[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h])); ui[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h] + P));
            if constexpr (EPI == EPI_XPAY_DOT) { vr[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h])); vi[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h] + P)); }
        }
        XOps xo[NOUT];
#pragma unroll
        for (int j = 0; j < NOUT; ++j) if (qi + j < q1) fetchXg(xo[j], qi + j);
        uint32_t const qFast = (q1 - qi > NOUT) ? qi + NOUT : q1;
        uint32_t const un = next_unit(u + NW);
        uint32_t const qNext = (un < nb) ? starts[first + un] : lastValid;
        if (q1 > q0) lastValid = q1 - 1;
        auto xof = [&](XOps& x, uint32_t q) __attribute__((always_inline)) {
            if (q < qi) {
                double const* Xb = L0 + size_t(pairs[2 * size_t(q) + 1] - first) * 2 * P;
#pragma unroll
                for (int h = 0; h < 2; ++h) { x.xr[h] = *(d2v const*)(Xb + h * 128); x.xi[h] = *(d2v const*)(Xb + P + h * 128); }
            } else {
                uint32_t const j = q - qi;
#pragma unroll
                for (int jj = 0; jj < NOUT; ++jj) if (j == uint32_t(jj)) x = xo[jj];
            }
        };
        auto mma = [&](AOps const& A, XOps const& x) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    R const nai = -A.ai[h][e];
                    cre = Acc<R>::mma(A.ar[h][e], x.xr[h][e], cre);
                    cim = Acc<R>::mma(A.ar[h][e], x.xi[h][e], cim);
                    cre = Acc<R>::mma(nai, x.xi[h][e], cre);
                    cim = Acc<R>::mma(A.ai[h][e], x.xr[h][e], cim);
                }
        };
        uint32_t q = q0;
        for (; q + 2 <= qFast; q += 2) {
            fetchA(Qa, q + 1);
            { XOps x; xof(x, q); mma(Pa, x); }
            __builtin_amdgcn_sched_barrier(0);
            fetchA(Pa, (q + 2 < qFast) ? q + 2 : qNext);
            { XOps x; xof(x, q + 1); mma(Qa, x); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < qFast) {
            fetchA(Qa, qNext);
            { XOps x; xof(x, q); mma(Pa, x); }
            __builtin_amdgcn_sched_barrier(0);
            Pa = Qa;
            ++q;
        }
        for (; q < q1; ++q) { AOps As; XOps x; fetchA(As, q); fetchXg(x, q); mma(As, x); }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2v yr, yi, nr, ni;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                yr[e] = cre[2 * h + e]; yi[e] = cim[2 * h + e];
                if constexpr (EPI == EPI_XPAY_DOT) {
                    R const tr = vr[h][e] + sr * ur[h][e] - si * ui[h][e], ti = vi[h][e] + si * ur[h][e] + sr * ui[h][e];
                    nr[e] = yr[e] + sr * tr - si * ti; ni[e] = yi[e] + si * tr + sr * ti;
                } else {
                    nr[e] = sr * yr[e] - si * yi[e] + ur[h][e]; ni[e] = si * yr[e] + sr * yi[e] + ui[h][e];
                }
                int const row = 2 * (lr + 4 * h) + e;
                double const wr = shadow_value(key, uint32_t(row * LN + lc)), wi = shadow_value(key, uint32_t(P + row * LN + lc));
                double const dr = nr[e], di = ni[e];
                part[0] += dr * wr - di * wi;
                part[1] += dr * wi + di * wr;
                if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] += dr * dr + di * di;
            }
            __builtin_nontemporal_store(yr, (d2v*)((R*)a.Y + eoff[h])); __builtin_nontemporal_store(yi, (d2v*)((R*)a.Y + eoff[h] + P));
            __builtin_nontemporal_store(nr, (d2v*)((R*)a.e0 + eoff[h])); __builtin_nontemporal_store(ni, (d2v*)((R*)a.e0 + eoff[h] + P));
        }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        double v = part[p];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) red[(wave * 3 + p) * LN + lane] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NPL * LN; e += NW * 64) {
        int const p = e / LN, j = e % LN;
        double sum = 0;
        for (int w = 0; w < NW; ++w) sum += red[(w * 3 + p) * LN + j];
        write_record<EPI>(a, chunk, LN, p, j, sum);
    }
}

// ---------------------------------------------------------------------------------------------------
// V8 = V6 (no LDS, interleaved layout) with the operands of ALL pairs of the wave as one stream, requested exactly one
// pair ahead without conditions (clamped at the end), so that the compiler's waits are exact counts.
template <int EPI, bool HASH, int WPS>
__global__ __launch_bounds__(256, WPS) void k_spmm_v8(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LM = 16, LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const nb = last - first;
    R const sr = (EPI == EPI_NONE) ? 0 : ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc], si = (EPI == EPI_NONE) ? 0 : ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc];
    double part[NPL > 0 ? NPL : 1] = {};
    __shared__ double s[4][3][LN];

    struct Ops { d2v ar[2], ai[2], xr[2], xi[2]; };
    R const* const A0 = (R const*)a.A + (lr * 16 + rowp(lc)) * 2;
    R const* const X0 = (R const*)a.X + (lr * 16 + lc) * 2;
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * P;
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            o.ar[h] = *(d2v const*)(Ab + h * 128); o.ai[h] = *(d2v const*)(Ab + P + h * 128);
            o.xr[h] = *(d2v const*)(Xb + h * 128); o.xi[h] = *(d2v const*)(Xb + P + h * 128);
        }
    };
    auto next_unit = [&](uint32_t u) -> uint32_t {
        while (u < nb && starts[first + u] == starts[first + u + 1]) u += 4;
        return u;
    };
    Ops Po, Qo;
    uint32_t lastValid = 0;
    {
        uint32_t const u0 = next_unit(wave);
        lastValid = (u0 < nb) ? starts[first + u0] : 0;
        fetch(Po, lastValid);
    }
    for (uint32_t u = wave; u < nb; u += 4) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    R const nai = -o.ai[h][e];
                    cre = Acc<R>::mma(o.ar[h][e], o.xr[h][e], cre);
                    cim = Acc<R>::mma(o.ar[h][e], o.xi[h][e], cim);
                    cre = Acc<R>::mma(nai, o.xi[h][e], cre);
                    cim = Acc<R>::mma(o.ai[h][e], o.xr[h][e], cim);
                }
        };
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        uint32_t const un = next_unit(u + 4);
        uint32_t const qNext = (un < nb) ? starts[first + un] : lastValid;
        if (q1 > q0) lastValid = q1 - 1;
        size_t const eoff[2] = { size_t(y) * 2 * P + (lr * 16 + lc) * 2, size_t(y) * 2 * P + ((lr + 4) * 16 + lc) * 2 };
        d2v ur[2], ui[2], vr[2], vi[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) if constexpr (EPI != EPI_NONE) {
            ur[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h])); ui[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + eoff[h] + P));
            if constexpr (EPI == EPI_XPAY_DOT) { vr[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h])); vi[h] = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + eoff[h] + P)); }
        }
        uint32_t q = q0;
        for (; q + 2 <= q1; q += 2) {
            fetch(Qo, q + 1);
            mma(Po);
            __builtin_amdgcn_sched_barrier(0);
            fetch(Po, (q + 2 < q1) ? q + 2 : qNext);
            mma(Qo);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (q < q1) {
            fetch(Qo, qNext);
            mma(Po);
            __builtin_amdgcn_sched_barrier(0);
            Po = Qo;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2v yr, yi, nr, ni;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                yr[e] = cre[2 * h + e]; yi[e] = cim[2 * h + e];
                if constexpr (EPI == EPI_NONE) continue;
                if constexpr (EPI == EPI_XPAY_DOT) {
                    R const tr = vr[h][e] + sr * ur[h][e] - si * ui[h][e], ti = vi[h][e] + si * ur[h][e] + sr * ui[h][e];
                    nr[e] = yr[e] + sr * tr - si * ti; ni[e] = yi[e] + si * tr + sr * ti;
                } else {
                    nr[e] = sr * yr[e] - si * yi[e] + ur[h][e]; ni[e] = si * yr[e] + sr * yi[e] + ui[h][e];
                }
                int const row = 2 * (lr + 4 * h) + e;
                double const wr = shadow_value(key, uint32_t(row * LN + lc)), wi = shadow_value(key, uint32_t(P + row * LN + lc));
                double const dr = nr[e], di = ni[e];
                part[0] += dr * wr - di * wi;
                part[1] += dr * wi + di * wr;
                if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] += dr * dr + di * di;
            }
            __builtin_nontemporal_store(yr, (d2v*)((R*)a.Y + eoff[h])); __builtin_nontemporal_store(yi, (d2v*)((R*)a.Y + eoff[h] + P));
            if constexpr (EPI != EPI_NONE) { __builtin_nontemporal_store(nr, (d2v*)((R*)a.e0 + eoff[h])); __builtin_nontemporal_store(ni, (d2v*)((R*)a.e0 + eoff[h] + P)); }
        }
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        double v = part[p];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (lane < 16) s[wave][p][lane] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NPL * LN; e += 256) {
        int const p = e / LN, j = e % LN;
        double const sum = ((s[0][p][j] + s[1][p][j]) + s[2][p][j]) + s[3][p][j];
        write_record<EPI>(a, chunk, LN, p, j, sum);
    }
}

// native <-> interleaved (one thread per element); kind 0: X-shaped, 1: A (rows permuted)
__global__ void k_relayout(double* dst, double const* src, size_t nBlocks, int kind, int toInterleaved) {
    size_t const n = nBlocks * 512;
    for (size_t t = blockIdx.x * size_t(blockDim.x) + threadIdx.x; t < n; t += size_t(gridDim.x) * blockDim.x) {
        size_t const b = t / 512; int const e = int(t % 512), c = e / 256, r = (e % 256) / 16, q = e % 16;   // native [c][r][q]
        int inter;
        if (kind == 0) inter = c * 256 + ((r / 2) * 16 + q) * 2 + (r % 2);               // X: r = row i, q = column j
        else inter = c * 256 + ((r / 2) * 16 + q) * 2 + (r % 2);                             // A native [k = r][i = q] -> [k/2][i][k%2]; the kernel looks up row ROWP(lane)
        if (toInterleaved) dst[b * 512 + inter] = src[t]; else dst[t] = src[b * 512 + inter];
    }
}

} // namespace tfq

// ---------------------------------------------------------------------------------------------------
using namespace tfq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(3); } } while (0)

template <typename T> std::vector<T> slurp(std::string const& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(1); }
    size_t const n = size_t(f.tellg()); f.seekg(0);
    std::vector<T> v(n / sizeof(T)); f.read((char*)v.data(), n); return v;
}
template <typename T> T* up(std::vector<T> const& v) { T* d; CK(hipMalloc((void**)&d, std::max<size_t>(v.size() * sizeof(T), 256))); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

__global__ void k_fill(double* p, size_t n, uint64_t seed, double scale) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        p[i] = scale * (double(splitmix64(seed + i) >> 11) * (1.0 / 9007199254740992.0) - 0.5);
}

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: %s <plan dir> <epi 1|2> <variant,variant,...> [reps]\n", argv[0]); return 1; }
    std::string const dir = argv[1];
    int const epi = atoi(argv[2]);
    std::string variants = argv[3];
    int const reps = argc > 4 ? atoi(argv[4]) : 20;
    size_t nX, nA, nP, nChunks, nCols; int LM, LN;
    { std::ifstream m(dir + "/meta.txt"); m >> nX >> nA >> nP >> nChunks >> nCols >> LM >> LN; }
    if (LM != 16 || LN != 16) { printf("lab is for 16x16 blocks\n"); return 1; }
    auto starts = slurp<uint32_t>(dir + "/starts.bin"), pairs = slurp<uint32_t>(dir + "/pairs.bin"), nIn = slurp<uint32_t>(dir + "/nInside.bin");
    auto chunkFirst = slurp<uint32_t>(dir + "/chunkFirst.bin"), chunkCol = slurp<uint32_t>(dir + "/chunkCol.bin"), order = slurp<uint32_t>(dir + "/order.bin");
    auto rowI = slurp<uint32_t>(dir + "/rowI.bin"); auto origCol = slurp<int32_t>(dir + "/origCol.bin");
    // host-side checks of what the kernels assume (a wrong index is a GPU fault)
    if (starts.size() != nX + 1 || pairs.size() != 2 * nP || chunkFirst.size() != nChunks + 1 || starts[nX] != nP || chunkFirst[nChunks] != nX) { printf("inconsistent plan\n"); return 1; }
    uint32_t maxChunk = 0;
    for (size_t c = 0; c < nChunks; ++c) { if (chunkFirst[c + 1] <= chunkFirst[c] || chunkCol[c] >= nCols || order[c] >= nChunks) { printf("bad chunk table\n"); return 1; } maxChunk = std::max(maxChunk, chunkFirst[c + 1] - chunkFirst[c]); }
    for (size_t p = 0; p < nP; ++p) if (pairs[2 * p] >= nA || pairs[2 * p + 1] >= nX) { printf("bad pair\n"); return 1; }
    {   // in-chunk pairs first, as counted
        std::vector<uint32_t> chunkOf(nX);
        for (size_t c = 0; c < nChunks; ++c) for (uint32_t b = chunkFirst[c]; b < chunkFirst[c + 1]; ++b) chunkOf[b] = uint32_t(c);
        for (size_t y = 0; y < nX; ++y) {
            uint32_t const lo = chunkFirst[chunkOf[y]], hi = chunkFirst[chunkOf[y] + 1];
            for (uint32_t q = starts[y]; q < starts[y + 1]; ++q) {
                bool const in = pairs[2 * q + 1] >= lo && pairs[2 * q + 1] < hi;
                if (in != (q < starts[y] + nIn[y])) { printf("nInside does not describe block %zu\n", y); return 1; }
            }
        }
    }
    printf("# %s: nX %zu nA %zu pairs %zu chunks %zu (max %u blocks) cols %zu\n", dir.c_str(), nX, nA, nP, nChunks, maxChunk, nCols);

    size_t const E = 512, S = nX * E;
    double *A, *v4, *v4init, *v6, *v8, *v9, *sc, *pz, *pd, *ref9, *ref4, *refpz; float* v3; Ctl* ctl;
    CK(hipMalloc((void**)&A, nA * E * 8)); CK(hipMalloc((void**)&v4, S * 8)); CK(hipMalloc((void**)&v4init, S * 8)); CK(hipMalloc((void**)&v6, S * 8));
    CK(hipMalloc((void**)&v8, S * 8)); CK(hipMalloc((void**)&v9, S * 8)); CK(hipMalloc((void**)&ref9, S * 8)); CK(hipMalloc((void**)&ref4, S * 8));
    CK(hipMalloc((void**)&v3, S * 4)); CK(hipMalloc((void**)&sc, nCols * 32 * 8)); CK(hipMalloc((void**)&pz, nChunks * 32 * 8)); CK(hipMalloc((void**)&refpz, nChunks * 32 * 8));
    CK(hipMalloc((void**)&pd, nChunks * 16 * 8)); CK(hipMalloc((void**)&ctl, 4096)); CK(hipMemset(ctl, 0, 4096)); CK(hipMemset(v3, 0, S * 4));
    k_fill<<<4096, 256>>>(A, nA * E, 1, 0.2); k_fill<<<4096, 256>>>(v4init, S, 2, 1.0); k_fill<<<4096, 256>>>(v6, S, 3, 1.0);
    k_fill<<<4096, 256>>>(v8, S, 4, 1.0); k_fill<<<64, 256>>>(sc, nCols * 32, 5, 0.6);
    CK(hipDeviceSynchronize());

    SpmmArgs a{};
    a.A = A; a.X = v6; a.Y = v9; a.e0 = v4; a.e1 = v8; a.sc = sc; a.v3 = v3; a.pz = pz; a.pd = pd; a.ctl = ctl; a.gate = 1;
    a.starts = up(starts); a.pairs = up(pairs); a.nY = uint32_t(nX); a.chunkFirst = up(chunkFirst); a.chunkCol = up(chunkCol); a.order = up(order);
    a.hashV3 = 1; a.origCol = up(origCol); a.rowI = up(rowI);
    uint32_t const* dIn = up(nIn);

    double *Ai, *v4i, *v6i, *v8i, *v9i;
    CK(hipMalloc((void**)&Ai, nA * E * 8)); CK(hipMalloc((void**)&v4i, S * 8)); CK(hipMalloc((void**)&v6i, S * 8)); CK(hipMalloc((void**)&v8i, S * 8)); CK(hipMalloc((void**)&v9i, S * 8));
    k_relayout<<<4096, 256>>>(Ai, A, nA, 1, 1); k_relayout<<<4096, 256>>>(v6i, v6, nX, 0, 1); k_relayout<<<4096, 256>>>(v8i, v8, nX, 0, 1);
    CK(hipDeviceSynchronize());
    SpmmArgs ai = a; ai.A = Ai; ai.X = v6i; ai.Y = v9i; ai.e0 = v4i; ai.e1 = v8i;
    auto launch = [&](std::string const& v) {
        dim3 const g{uint32_t(nChunks)}, b{256};
        if (v == "v0") { if (epi == 0) k_spmm_mfma<double, 16, 16, EPI_NONE, false, false, false><<<g, b>>>(a); else if (epi == 1) k_spmm_mfma<double, 16, 16, EPI_XPAY_DOT, true, false, true><<<g, b>>>(a); else k_spmm_mfma<double, 16, 16, EPI_AXPY_NRM_DOT, true, false, true><<<g, b>>>(a); }
        else if (v == "v2") { if (maxChunk > 16) { printf("v2 needs chunks of at most 16 blocks\n"); exit(1); }
                              if (epi == 1) k_spmm_ldsx<EPI_XPAY_DOT, true, 16, 4, 0><<<g, b>>>(a, dIn); else k_spmm_ldsx<EPI_AXPY_NRM_DOT, true, 16, 4, 0><<<g, b>>>(a, dIn); }
        else if (v == "v3") { if (maxChunk > 16) { printf("needs chunks of at most 16 blocks\n"); exit(1); } if (epi == 1) k_spmm_v3<EPI_XPAY_DOT, true, 16, 2, 4><<<g, b>>>(a, dIn); else k_spmm_v3<EPI_AXPY_NRM_DOT, true, 16, 2, 4><<<g, b>>>(a, dIn); }
        else if (v == "v4") { if (maxChunk > 16) { printf("needs chunks of at most 16 blocks\n"); exit(1); } if (epi == 1) k_spmm_v4<EPI_XPAY_DOT, true, 16, 2, 4><<<g, b>>>(a, dIn); else k_spmm_v4<EPI_AXPY_NRM_DOT, true, 16, 2, 4><<<g, b>>>(a, dIn); }
        else if (v == "v6") { if (epi == 0) k_spmm_v6<EPI_NONE, false><<<g, b>>>(ai); else if (epi == 1) k_spmm_v6<EPI_XPAY_DOT, true><<<g, b>>>(ai); else k_spmm_v6<EPI_AXPY_NRM_DOT, true><<<g, b>>>(ai); }
        else if (v == "v7") { if (maxChunk > 16) { printf("needs chunks of at most 16 blocks\n"); exit(1); } if (epi == 1) k_spmm_v7<EPI_XPAY_DOT, true, 16, 2, 4><<<g, b>>>(ai, dIn); else k_spmm_v7<EPI_AXPY_NRM_DOT, true, 16, 2, 4><<<g, b>>>(ai, dIn); }
        else if (v == "v7c12") { if (maxChunk > 12) { printf("needs chunks of at most 12 blocks\n"); exit(1); } if (epi == 1) k_spmm_v7<EPI_XPAY_DOT, true, 12, 2, 4, 3><<<g, b>>>(ai, dIn); else k_spmm_v7<EPI_AXPY_NRM_DOT, true, 12, 2, 4, 3><<<g, b>>>(ai, dIn); }
        else if (v == "v6nt") { if (epi == 0) k_spmm_v6nt<EPI_NONE, false, 1><<<g, b>>>(ai); else if (epi == 1) k_spmm_v6nt<EPI_XPAY_DOT, true, 1><<<g, b>>>(ai); else k_spmm_v6nt<EPI_AXPY_NRM_DOT, true, 1><<<g, b>>>(ai); }
        else if (v == "v6nt2") { if (epi == 0) k_spmm_v6nt<EPI_NONE, false, 2><<<g, b>>>(ai); else if (epi == 1) k_spmm_v6nt<EPI_XPAY_DOT, true, 2><<<g, b>>>(ai); else k_spmm_v6nt<EPI_AXPY_NRM_DOT, true, 2><<<g, b>>>(ai); }
        else if (v == "v8") { if (epi == 0) k_spmm_v8<EPI_NONE, false, 3><<<g, b>>>(ai); else if (epi == 1) k_spmm_v8<EPI_XPAY_DOT, true, 3><<<g, b>>>(ai); else k_spmm_v8<EPI_AXPY_NRM_DOT, true, 3><<<g, b>>>(ai); }
        else if (v == "v8w4") { if (epi == 1) k_spmm_v8<EPI_XPAY_DOT, true, 4><<<g, b>>>(ai); else k_spmm_v8<EPI_AXPY_NRM_DOT, true, 4><<<g, b>>>(ai); }
        else if (v == "v5") { if (maxChunk > 16) { printf("needs chunks of at most 16 blocks\n"); exit(1); } if (epi == 1) k_spmm_v5<EPI_XPAY_DOT, true, 16, 4, 2><<<g, b>>>(a, dIn); else k_spmm_v5<EPI_AXPY_NRM_DOT, true, 16, 4, 2><<<g, b>>>(a, dIn); }
        else if (v == "v5c12") { if (maxChunk > 12) { printf("needs chunks of at most 12 blocks\n"); exit(1); } if (epi == 1) k_spmm_v5<EPI_XPAY_DOT, true, 12, 4, 1, 3><<<g, b>>>(a, dIn); else k_spmm_v5<EPI_AXPY_NRM_DOT, true, 12, 4, 1, 3><<<g, b>>>(a, dIn); }
        else if (v == "v5c12u3") { if (maxChunk > 12) { printf("needs chunks of at most 12 blocks\n"); exit(1); } if (epi == 1) k_spmm_v5<EPI_XPAY_DOT, true, 12, 4, 3, 3><<<g, b>>>(a, dIn); else k_spmm_v5<EPI_AXPY_NRM_DOT, true, 12, 4, 3, 3><<<g, b>>>(a, dIn); }
        else if (v == "v5c8") { if (maxChunk > 8) { printf("needs chunks of at most 8 blocks\n"); exit(1); } if (epi == 1) k_spmm_v5<EPI_XPAY_DOT, true, 8, 4, 1, 4><<<g, b>>>(a, dIn); else k_spmm_v5<EPI_AXPY_NRM_DOT, true, 8, 4, 1, 4><<<g, b>>>(a, dIn); }
        else if (v == "v5c8u2") { if (maxChunk > 8) { printf("needs chunks of at most 8 blocks\n"); exit(1); } if (epi == 1) k_spmm_v5<EPI_XPAY_DOT, true, 8, 4, 2, 4><<<g, b>>>(a, dIn); else k_spmm_v5<EPI_AXPY_NRM_DOT, true, 8, 4, 2, 4><<<g, b>>>(a, dIn); }
        else if (v == "v5u1") { if (maxChunk > 16) { printf("needs chunks of at most 16 blocks\n"); exit(1); } if (epi == 1) k_spmm_v5<EPI_XPAY_DOT, true, 16, 4, 1><<<g, b>>>(a, dIn); else k_spmm_v5<EPI_AXPY_NRM_DOT, true, 16, 4, 1><<<g, b>>>(a, dIn); }
        else if (v == "v3c12") { if (maxChunk > 12) { printf("needs chunks of at most 12 blocks\n"); exit(1); }
                                 if (epi == 1) k_spmm_v3<EPI_XPAY_DOT, true, 12, 2, 4, 3><<<g, b>>>(a, dIn); else k_spmm_v3<EPI_AXPY_NRM_DOT, true, 12, 2, 4, 3><<<g, b>>>(a, dIn); }
        else if (v == "v3c8") { if (maxChunk > 8) { printf("needs chunks of at most 8 blocks\n"); exit(1); }
                                 if (epi == 1) k_spmm_v3<EPI_XPAY_DOT, true, 8, 2, 4, 4><<<g, b>>>(a, dIn); else k_spmm_v3<EPI_AXPY_NRM_DOT, true, 8, 2, 4, 4><<<g, b>>>(a, dIn); }
        else if (v == "v2a1") k_spmm_ldsx<EPI_XPAY_DOT, true, 16, 4, 1><<<g, b>>>(a, dIn);
        else if (v == "v2a2") k_spmm_ldsx<EPI_XPAY_DOT, true, 16, 4, 2><<<g, b>>>(a, dIn);
        else if (v == "v2w8") { if (epi == 1) k_spmm_ldsx<EPI_XPAY_DOT, true, 16, 8, 0><<<g, dim3(512)>>>(a, dIn); else k_spmm_ldsx<EPI_AXPY_NRM_DOT, true, 16, 8, 0><<<g, dim3(512)>>>(a, dIn); }
        else if (v == "v2w8a1") k_spmm_ldsx<EPI_XPAY_DOT, true, 16, 8, 1><<<g, dim3(512)>>>(a, dIn);
        else if (v == "v2w8a2") k_spmm_ldsx<EPI_XPAY_DOT, true, 16, 8, 2><<<g, dim3(512)>>>(a, dIn);
        else if (v == "v2b") { if (maxChunk > 16) { printf("v2b needs chunks of at most 16 blocks\n"); exit(1); }
                              if (epi == 1) k_spmm_ldsx_b<EPI_XPAY_DOT, true, 16, 2><<<g, b>>>(a, dIn); else k_spmm_ldsx_b<EPI_AXPY_NRM_DOT, true, 16, 2><<<g, b>>>(a, dIn); }
        else { printf("unknown variant %s\n", v.c_str()); exit(1); }
    };
    // reference results: the shipped kernel on the same lists
    CK(hipMemcpy(v4, v4init, S * 8, hipMemcpyDeviceToDevice)); launch("v0"); CK(hipDeviceSynchronize()); CK(hipGetLastError());
    CK(hipMemcpy(ref9, v9, S * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(ref4, v4, S * 8, hipMemcpyDeviceToDevice)); CK(hipMemcpy(refpz, pz, nChunks * 32 * 8, hipMemcpyDeviceToDevice));
    std::vector<double> h9(S), r9(S), hz(nChunks * 32), rz(nChunks * 32);
    CK(hipMemcpy(r9.data(), ref9, S * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rz.data(), refpz, nChunks * 32 * 8, hipMemcpyDeviceToHost));

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    size_t pos = 0;
    while (pos < variants.size()) {
        size_t const c = variants.find(',', pos);
        std::string const v = variants.substr(pos, c == std::string::npos ? std::string::npos : c - pos);
        pos = (c == std::string::npos) ? variants.size() : c + 1;
        CK(hipMemcpy(v4, v4init, S * 8, hipMemcpyDeviceToDevice)); CK(hipMemset(v9, 0, S * 8)); CK(hipMemset(pz, 0, nChunks * 32 * 8));
        if (v[1] >= '6') { k_relayout<<<4096, 256>>>(v4i, v4init, nX, 0, 1); CK(hipMemset(v9i, 0, S * 8)); }
        launch(v); CK(hipDeviceSynchronize()); CK(hipGetLastError());
        if (v[1] >= '6') { k_relayout<<<4096, 256>>>(v9, v9i, nX, 0, 0); CK(hipDeviceSynchronize()); }
        CK(hipMemcpy(h9.data(), v9, S * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hz.data(), pz, nChunks * 32 * 8, hipMemcpyDeviceToHost));
        double d9 = 0, m9 = 0, dz = 0, mz = 0;
        for (size_t i = 0; i < S; ++i) { d9 = std::max(d9, std::abs(h9[i] - r9[i])); m9 = std::max(m9, std::abs(r9[i])); }
        for (size_t i = 0; i < hz.size(); ++i) { dz = std::max(dz, std::abs(hz[i] - rz[i])); mz = std::max(mz, std::abs(rz[i])); }
        for (int w = 0; w < 3; ++w) launch(v);
        CK(hipDeviceSynchronize());
        std::vector<float> ts;
        for (int s = 0; s < 5; ++s) {
            CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) launch(v); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / reps);
        }
        std::sort(ts.begin(), ts.end());
        double const bytes = (epi == 0 ? 2.0 * S * 8 : (epi == 1 ? 5.0 : 4.0) * S * 8 + S * 4) + double(nA) * 4096 + 4.0 * (nX + 1) + 8.0 * nP;
        printf("%-6s epi %d: min %.4f ms median %.4f ms  (%.0f GB/s algorithmic, frac %.3f of 8 TB/s) | max|dY| %.2e of %.2e, max|dpz| %.2e of %.2e\n",
               v.c_str(), epi, ts[0], ts[2], bytes / ts[0] * 1e-6, bytes / ts[0] * 1e-6 / 8000, d9, m9, dz, mz);
        fflush(stdout);
    }
    return 0;
}
