// Block-sparse multiply Y = A*X for gfx950 (MI355X), with the vector updates that follow it in
// the tfQMR iteration fused into the epilogue.
//
// Contract (same as the reference kernel gemmNxNf, real-space/tfQMRgpu tfqmrgpu_blockmult.hxx:9-93
// and its launcher tfqmrgpu_blocksparse.hxx:71-199):
//     Y[iY][c][i][j] = sum_{p in starts[iY]..starts[iY+1]} sum_k A[pairs[2p]][c'][k][i] * X[pairs[2p+1]][c''][k][j]
// complex arithmetic on split Re/Im planes, A blocks stored transposed ([k][i]), accumulation in the
// storage precision.  Flop count nPairs*8*LM*LM*LN (tfqmrgpu_blocksparse.hxx:198).
//
// Implementations -- first the four of the plans that keep groups of rows interleaved (tfq_device.hpp: ilv_offset; every operand,
// epilogue operand and result of a lane is one 16-byte access), the hot shapes of the BASELINE configurations:
//  * k_spmm_ilv16  : 16 x 16 complex<double>, row pairs (configs 2 and 4);   k_spmm_ilv8 : 8 x 8 complex<double>, a block = one access (config 5);
//  * k_spmm_ilv16f : 16 x 16 complex<float>, row quads;                      k_spmm_ilvf : 16 | 32 x 32 complex<float>, row quads (32 x 32: config 3);
//  * k_spmm_ilv8w  : 8 x 32 | 64 complex<double>, row pairs (r03);            k_spmm_ilv8f : 8 x 8 | 32 | 64 complex<float>, row quads, two products per tile (r03);
// then, on the reference's native order (every other shape, caller-owned arrays of tfqmrgpuExt_multiply, TFQMRGPU_ILV=0):
//  * k_spmm_mfma : LM and LN multiples of 16.  One wavefront owns a 16 x LN strip of one Y block
//    and keeps it in MFMA accumulators (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32).  The
//    native layouts ARE the MFMA operand layouts: lane l feeds A[k0 + l/16][i0 + l%16] and
//    X[k0 + l/16][columns of lane l%16], i.e. four consecutive rows per load instruction, so operands go
//    global -> VGPR fully coalesced with no LDS transpose.  With several 16-column tiles per strip a lane
//    owns NEIGHBOURING columns (ColMap below) and moves them as one 16-byte access.  4 real MFMA chains
//    per complex product (-Im(A) is formed once per operand), 3 in double above 16 x 16 (Slice::mma3); the 16 x 16
//    instances recompute the shadow vector from its hash instead of reading it (HASH).
//  * k_spmm_mfma8 : LM == 8 (and 4 x 32 z): [Re A; Im A] x [Re X | Im X] fills one 16 x 16 tile per 8 block columns (LN = 9,
//    10: the last tile is masked; LM == 4: half of the rows are empty -- the matrix pipe is idle in these HBM-bound shapes).
//  * k_spmm_m4     : 4 x 4 | 8 | 32 in double: four 4 x 4 x 4 products per v_mfma_f64_4x4x4_4b_f64, elements straight from the planes, no LDS.
//  * k_spmm_s4w    : 4 x 8 | 32 in float (4 x 4 without epilogue): k_spmm_small4's arithmetic with two | four neighbouring columns per lane.
//  * k_spmm_small4 : the other 4-row shapes: one lane per element, operands once per thread group through LDS.
//  * k_spmm_direct : one thread per output element; only as the epilogue of a user-defined operator,
//    operands through the vector L1.
// A work group processes one chunk (run of Y blocks of one block column, tfq_plan.cpp), so the
// per-RHS scalars of the epilogue are uniform and the dot / norm contributions leave the work group
// as one [LN] record (deterministic order).
#include <cstdlib>
#include <type_traits>

#include "tfq_device.hpp"
#include "tfq_vec.hpp"
#include "tfq_switch.hpp"
#include "tfq_colops.hpp"

// (timing-only variants of the kernels below -- block products or fetches skipped, stamps, the shader clock under load: results wrong by
//  construction -- live in a copy of their own, scripts/lab/tfq_spmm_probes.hip, built with scripts/build_variant.sh; nothing of them is in this file)

namespace tfq {

struct SpmmArgs {
    void* Y; void const* A; void const* X;
    uint32_t const* starts; uint32_t const* pairs;
    uint32_t nY;                       // number of Y blocks (plain mode)
    uint32_t const* chunkFirst;        // nullptr: plain mode, chunk b = blocks [b*CH, (b+1)*CH)
    uint32_t const* chunkCol;
    uint32_t const* order;             // launch order: work group b processes chunk order[b] (nullptr: b)
    uint32_t CH;
    Ctl const* ctl; int gate;          // 0: always run, 1: skip when the solve has stopped, 2: only when probing
    void* e0; void const* e1; void const* sc; float const* v3;
    void const* B; uint32_t const* bOfX;       // bOfX == nullptr: B is a whole X-shaped vector (block y of B belongs to Y block y: the
                                               // residual of the mixed-precision refinement as the right-hand side, DevPlan::R)
    double* pz; double* pd;
    void const* Yext; uint32_t const* yPerm;   // k_spmm_direct only: take block y of the product from Yext[yPerm[y]]
    int hashV3;                        // the shadow vector is the counter-based hash (tfq_device.hpp): recompute it, do not read it
    int32_t const* origCol; uint32_t const* rowI;   // original block column per compressed column, block row per Y block
    int ilv;                           // element order of the plan's blocks (tfq_device.hpp: ilv_offset); the plain mode is always native
    int aOnce;                         // every A block is used about once per multiply (few block columns): stream A past the caches
    int first;                         // EPI_XPAY_DOT in the first iteration of a solve: old v4 = v8 = 0 by definition, not read (DevPlan::first)
    int m3;                            // double shapes above 16 x 16: three real products per complex one (tfqmrgpuExt_setThreeProductMultiply)
    DevPlan const* foldPlan;           // not null: the column operation that consumes this launch's records runs in its tail (tfq_colops.hpp)
    uint8_t const* colBatch; uint32_t const* colStart; uint32_t const* colChunkPtr;   // k_spmm_ilv8b: (batch size << 4) | position per block column; block / chunk ranges of the columns
    uint32_t const* yOrder;            // plain mode, not null: a prepared order (tfq_order.cpp) -- position i of the launch computes Y block yOrder[i]
    uint32_t plainPer;                 // plain mode of k_spmm_mfma, not 0: XCD x (work groups x, x + 8, ...) takes the chunks [x * plainPer, (x + 1) * plainPer)
};

// data that a kernel touches once (epilogue vectors) moves non-temporally, so that the stream does not push the A and
// X blocks, which neighbouring work groups re-use, out of the L2 (measured on P2: fused multiply 0.825 -> 0.777 ms)
// Only where a wave's access covers runs of at least 64 bytes: 32-byte runs (the 8-column tiles of k_spmm_mfma8 in
// float) as non-temporal partial writes cost 2x (8x32 `c`: 1.27 -> 2.79 ms), so STREAM is a template switch.
template <bool STREAM, typename T> __device__ inline T ld_stream(T const* p) { if constexpr (STREAM) return __builtin_nontemporal_load(p); else return *p; }
template <bool STREAM, typename T> __device__ inline void st_stream(T* p, T v) { if constexpr (STREAM) __builtin_nontemporal_store(v, p); else *p = v; }

template <int EPI> struct EpiPlanes { static constexpr int N = (EPI == EPI_XPAY_DOT) ? 2 : (EPI == EPI_AXPY_NRM_DOT) ? 3 : (EPI == EPI_RESIDUAL) ? 1 : 0; };

// Every epilogue writes its complex updates and reductions as EXPLICIT fused multiply-adds, the same pattern in every kernel: what the
// compiler contracts on its own changes with the code around an expression (a refactoring of the operand loads moved the last bits of
// the 4-row shapes, amplified to 6e-6 in the bound history of a 32-iteration solve), and the instances of one kernel that read or
// recompute the shadow vector must round alike (tests/test_gpu_hash_mode.py compares them bit by bit).
__device__ inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// v4 := v9 + s (v8 + s v4)   (u = old v4, x = v8, y = v9 = A v6; tfqmrgpu_core.hxx:196-202)
template <typename R> __device__ inline void epi_xpay2(R& nr, R& ni, R yr, R yi, R ur, R ui, R xr, R xi, R sr, R si) {
    R const tr = fma_(-si, ui, fma_(sr, ur, xr)), ti = fma_(sr, ui, fma_(si, ur, xi));
    nr = fma_(-si, ti, fma_(sr, tr, yr)); ni = fma_(sr, ti, fma_(si, tr, yi));
}
// v5 := s v8 + v5   (u = old v5, y = v8 = A v6; tfqmrgpu_core.hxx:224-228)
template <typename R> __device__ inline void epi_axpy(R& nr, R& ni, R yr, R yi, R ur, R ui, R sr, R si) {
    nr = fma_(-si, yi, fma_(sr, yr, ur)); ni = fma_(sr, yi, fma_(si, yr, ui));
}
// pz += v3 . d (unconjugated), pd += |d|^2, in double
__device__ inline void epi_dot(double& p0, double& p1, double dr, double di, double wr, double wi) {
    p0 = __builtin_fma(-di, wi, __builtin_fma(dr, wr, p0)); p1 = __builtin_fma(di, wr, __builtin_fma(dr, wi, p1));
}
__device__ inline void epi_nrm(double& p, double dr, double di) { p = __builtin_fma(di, di, __builtin_fma(dr, dr, p)); }

// per-element epilogue; off = offset of the element's real part in an X-shaped vector, P = plane size.
// In two steps so that a kernel can request the operands (old v4|v5, v8, v3) before its block products and use
// them behind: EpiElem::load, epilogue_apply; epilogue() is the two in a row.
template <typename R, int EPI, bool STREAM>
struct EpiElem {
    R ur, ui, xr, xi; float wr, wi;
    __device__ inline void load(SpmmArgs const& a, size_t off, int P) {
        if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
            wr = ld_stream<STREAM>(a.v3 + off); wi = ld_stream<STREAM>(a.v3 + off + P);
        }
        if constexpr (EPI == EPI_XPAY_DOT) if (a.first) { ur = 0; ui = 0; xr = 0; xi = 0; return; }
        if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
            R const* u = (R const*)a.e0;
            ur = ld_stream<STREAM>(u + off); ui = ld_stream<STREAM>(u + off + P);
        }
        if constexpr (EPI == EPI_XPAY_DOT) {
            R const* v8 = (R const*)a.e1;
            xr = ld_stream<STREAM>(v8 + off); xi = ld_stream<STREAM>(v8 + off + P);
        }
    }
};

template <typename R, int EPI, bool STREAM>
__device__ inline void epilogue_apply(SpmmArgs const& a, size_t off, int P, R yr, R yi, R sr, R si,
                                      EpiElem<R, EPI, STREAM> const& o, uint32_t bq, int eoff, double* acc /* [planes] */)
{
    if constexpr (EPI == EPI_NONE) {
        st_stream<STREAM>((R*)a.Y + off, yr); st_stream<STREAM>((R*)a.Y + off + P, yi);
    } else if constexpr (EPI == EPI_XPAY_DOT) {
        // v9 := A v6 (kept for the v5 update); v4 := v8 + beta v4; v4 := v9 + beta v4; pz += v3 . v4
        // (tfqmrgpu_core.hxx:196-202)
        st_stream<STREAM>((R*)a.Y + off, yr); st_stream<STREAM>((R*)a.Y + off + P, yi);
        R* v4 = (R*)a.e0;
        R ur, ui;
        epi_xpay2(ur, ui, yr, yi, o.ur, o.ui, o.xr, o.xi, sr, si);
        st_stream<STREAM>(v4 + off, ur); st_stream<STREAM>(v4 + off + P, ui);
        epi_dot(acc[0], acc[1], ur, ui, o.wr, o.wi);
    } else if constexpr (EPI == EPI_AXPY_NRM_DOT) {
        // v8 := A v6; v5 := alfa v8 + v5; pd += |v5|^2; pz += v3 . v5  (tfqmrgpu_core.hxx:224-228,189)
        st_stream<STREAM>((R*)a.Y + off, yr); st_stream<STREAM>((R*)a.Y + off + P, yi);
        R* v5 = (R*)a.e0;
        R nr, ni;
        epi_axpy(nr, ni, yr, yi, o.ur, o.ui, sr, si);
        st_stream<STREAM>(v5 + off, nr); st_stream<STREAM>(v5 + off + P, ni);
        epi_dot(acc[0], acc[1], nr, ni, o.wr, o.wi);
        epi_nrm(acc[2], nr, ni);
    } else { // EPI_RESIDUAL: |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
        R rr = yr, ri = yi;
        if (bq != 0xffffffffu) {
            R const* b = (R const*)a.B + size_t(bq) * 2 * P;
            rr += R(-1) * b[eoff]; ri += R(-1) * b[eoff + P];
        }
        epi_nrm(acc[0], rr, ri);
    }
}

template <typename R, int EPI, bool STREAM = true>
__device__ inline void epilogue(SpmmArgs const& a, size_t off, int P, R yr, R yi, R sr, R si,
                                uint32_t bq, int eoff, double* acc /* [planes] */)
{
    EpiElem<R, EPI, STREAM> o;
    o.load(a, off, P);
    epilogue_apply<R, EPI, STREAM>(a, off, P, yr, yi, sr, si, o, bq, eoff, acc);
}

template <int EPI>
__device__ inline void write_record(SpmmArgs const& a, uint32_t chunk, int LN, int p, int j, double v) {
    // (folded: the record is read by ANOTHER work group of this launch, the last one of the column -- coherent store, tfq_colops.hpp: co_store)
    bool const co = (a.foldPlan != nullptr);
    if constexpr (EPI == EPI_XPAY_DOT) st_record(a.pz + (size_t(chunk) * 2 + p) * LN + j, v, co);
    else if constexpr (EPI == EPI_AXPY_NRM_DOT) { if (p < 2) st_record(a.pz + (size_t(chunk) * 2 + p) * LN + j, v, co); else st_record(a.pd + size_t(chunk) * LN + j, v, co); }
    else if constexpr (EPI == EPI_RESIDUAL) st_record(a.pd + size_t(chunk) * LN + j, v, co);
}

// the column operation behind a fused multiply, run by the last work group of the column (small systems, tfq_colops.hpp)
template <typename R, int LN, int EPI>
__device__ inline void spmm_fold(SpmmArgs const& a, uint32_t col) {
    __shared__ ColScratch sc;
    constexpr int WHAT = (EPI == EPI_XPAY_DOT) ? FOLD_DEC34 : (EPI == EPI_AXPY_NRM_DOT) ? FOLD_DECT_FINAL : FOLD_PROBE;
    fold_tail<R, LN, WHAT>(*a.foldPlan, col, sc);
}

__device__ inline bool gate_closed(SpmmArgs const& a) {
    if (a.gate == 0) return false;
    if (a.ctl->state != 0) return true;
    return (a.gate == 2 && a.ctl->probe == 0);
}

// ---------------------------------------------------------------------------------------------------
// one thread per output element
template <typename R, int LM, int LN, int EPI>
__global__ __launch_bounds__(256) void k_spmm_direct(SpmmArgs a) {
    if (gate_closed(a)) return;
    constexpr int P = LM * LN;                       // elements per plane
    constexpr int NACC = (P >= 256) ? P / 256 : 1;   // outputs per thread
    constexpr int GRP = (P >= 256) ? 1 : 256 / P;    // Y blocks in flight per work group
    constexpr int NPL = EpiPlanes<EPI>::N;
    static_assert(P < 256 || P % 256 == 0, "block does not tile the work group");
    int const t = threadIdx.x;
    int const g = (P >= 256) ? 0 : t / P;
    int const e0 = (P >= 256) ? t : t % P;
    bool const active = (g < GRP);
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;   // XCD-aware launch order (tfq_plan.cpp)
    uint32_t first, last, col = 0;
    if (a.chunkFirst) { first = a.chunkFirst[chunk]; last = a.chunkFirst[chunk + 1]; col = a.chunkCol[chunk]; }
    else { first = chunk * a.CH; last = min(first + a.CH, a.nY); }

    int const j = e0 % LN;                           // 256 % LN == 0 whenever NACC > 1
    R sr = 0, si = 0;
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
        sr = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + j];
        si = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + j];
    }
    double part[NPL > 0 ? NPL : 1] = {};

    if (active) for (uint32_t y = first + g; y < last; y += GRP) {
        R yr[NACC], yi[NACC];
#pragma unroll
        for (int n = 0; n < NACC; ++n) { yr[n] = 0; yi[n] = 0; }
        if (a.Yext) {   // product computed by a user-defined operator
            R const* Yb = (R const*)a.Yext + size_t(a.yPerm[y]) * 2 * P;
#pragma unroll
            for (int n = 0; n < NACC; ++n) { yr[n] = Yb[e0 + n * 256]; yi[n] = Yb[P + e0 + n * 256]; }
        } else
        for (uint32_t q = a.starts[y]; q < a.starts[y + 1]; ++q) {
            R const* Ab = (R const*)a.A + size_t(a.pairs[2 * size_t(q)]) * 2 * LM * LM;
            R const* Xb = (R const*)a.X + size_t(a.pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                int const e = e0 + n * 256, i = e / LN;
                R cr = 0, ci = 0;
#pragma unroll 4
                for (int k = 0; k < LM; ++k) {
                    R const ar = Ab[k * LM + i], ai = Ab[LM * LM + k * LM + i];
                    R const xr = Xb[k * LN + j], xi = Xb[P + k * LN + j];
                    cr = fma_(-ai, xi, fma_(ar, xr, cr));
                    ci = fma_(ai, xr, fma_(ar, xi, ci));
                }
                yr[n] += cr; yi[n] += ci;
            }
        }
        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            int const e = e0 + n * 256;
            int const el = plane_offset(a.ilv, e / LN, e % LN, LN);      // where the plan keeps element (row, column)
            epilogue<R, EPI>(a, size_t(y) * 2 * P + el, P, yr[n], yi[n], sr, si, bq, el, part);
        }
    }

    if constexpr (NPL > 0) {
        // threads that share j: rank = position among them; sum in rank order
        constexpr int RANKS = (P >= 256) ? 256 / LN : GRP * LM;
        __shared__ double s[NPL * LN * RANKS];
        int const rank = (P >= 256) ? t / LN : g * LM + e0 / LN;
        if (active)
#pragma unroll
            for (int p = 0; p < NPL; ++p) s[(p * LN + j) * RANKS + rank] = part[p];
        __syncthreads();
        for (int e = t; e < NPL * LN; e += 256) {
            double sum = 0;
#pragma unroll 1
            for (int r = 0; r < RANKS; ++r) sum += s[e * RANKS + r];
            write_record<EPI>(a, chunk, LN, e / LN, e % LN, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// MFMA kernel: LM % 16 == 0, LN % 16 == 0
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;
template <typename R> struct Acc;
template <> struct Acc<double> {
    using T = d4;
    __device__ static inline T mma(double a, double b, T c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f64_16x16x4_f64: register r of lane l is row (l/16) + 4 r, column l%16
    __device__ static inline int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Acc<float> {
    using T = f4;
    __device__ static inline T mma(float a, float b, T c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    // C/D layout of v_mfma_f32_16x16x4_f32: register r of lane l is row 4 (l/16) + r, column l%16
    __device__ static inline int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};

// NT consecutive elements as one access (NT * sizeof(R) bytes, naturally aligned by construction)
template <typename R, int N> struct VecOf { using T = R __attribute__((ext_vector_type(N))); };
template <typename R, int N>
__device__ inline void vload(R (&dst)[N], R const* p) {
    if constexpr (N == 1) dst[0] = *p;
    else {
        auto const v = *reinterpret_cast<typename VecOf<R, N>::T const*>(p);
#pragma unroll
        for (int i = 0; i < N; ++i) dst[i] = v[i];
    }
}
template <typename R, int N>
__device__ inline void vstore(R* p, R const (&src)[N]) {
    if constexpr (N == 1) *p = src[0];
    else {
        typename VecOf<R, N>::T v;
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = src[i];
        *reinterpret_cast<typename VecOf<R, N>::T*>(p) = v;
    }
}
// the same for data that is touched once (epilogue vectors): non-temporal where STREAM (see ld_stream above)
template <bool STREAM, typename R, int N>
__device__ inline void vload_stream(R (&dst)[N], R const* p) {
    if constexpr (!STREAM) vload<R, N>(dst, p);
    else if constexpr (N == 1) dst[0] = __builtin_nontemporal_load(p);
    else {
        auto const v = __builtin_nontemporal_load(reinterpret_cast<typename VecOf<R, N>::T const*>(p));
#pragma unroll
        for (int i = 0; i < N; ++i) dst[i] = v[i];
    }
}
template <bool STREAM, typename R, int N>
__device__ inline void vstore_stream(R* p, R const (&src)[N]) {
    if constexpr (!STREAM) vstore<R, N>(p, src);
    else if constexpr (N == 1) __builtin_nontemporal_store(src[0], p);
    else {
        typename VecOf<R, N>::T v;
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = src[i];
        __builtin_nontemporal_store(v, reinterpret_cast<typename VecOf<R, N>::T*>(p));
    }
}

// Column map of the MFMA kernel: a lane touches NT block columns (one per accumulator tile).  They are chosen as
// NT/VW groups of VW NEIGHBOURS, VW * sizeof(R) = 16 bytes where NT allows: tile nt of lane column lc holds block
// column (nt/VW) * 16 VW + lc * VW + nt % VW (not nt * 16 + lc).  X operands and every epilogue vector then move
// as 16-byte accesses, 256 contiguous bytes per row and lane group (the memory pipe retires one wave-wide access
// per 16 clocks whatever its width, scripts/ta_rate.hip).  Which 16 columns share a tile is free.
template <typename R, int NT> struct ColMap {
    static constexpr int VW = (NT * sizeof(R) <= 16) ? NT : int(16 / sizeof(R));   // columns per access
    static constexpr int NG = NT / VW;                                             // accesses per row
    __device__ static inline int col(int lc, int nt) { return (nt / VW) * 16 * VW + lc * VW + nt % VW; }
};

// row tiles per wave: two where the block has them and the accumulators (MS * NT complex tiles) stay within 64 VGPRs
template <typename R, int MT, int NT> struct RowTiles {
    static constexpr int MS = (MT % 2 == 0 && 2 * NT * sizeof(R) <= 32) ? 2 : 1;
};

// operands of one "slice" = KSL consecutive MFMA k-steps (4 k each) of one block product, for a strip of
// MS * 16 block rows: the wave owns MS row tiles, tile ms of lane column lc holds block row i0 + lc * MS + ms, so
// that the A operand too moves as one MS-wide access and every X operand feeds MS tiles
template <typename R, int MS, int NT, int KSL>
struct Slice {
    R ar[KSL][MS], ai[KSL][MS], xr[KSL][NT], xi[KSL][NT];
    // Ab: A block + first row of this lane (i0 + lc * MS), Xb: X block + first column of this lane (lc * VW)
    template <int LM, int LN>
    __device__ inline void load(R const* __restrict__ Ab, R const* __restrict__ Xb, int k0, int lr) {
        constexpr int P = LM * LN, VW = ColMap<R, NT>::VW, NG = ColMap<R, NT>::NG;
#pragma unroll
        for (int s = 0; s < KSL; ++s) {
            int const k = k0 + 4 * s + lr;
            vload<R, MS>(ar[s], Ab + k * LM); vload<R, MS>(ai[s], Ab + LM * LM + k * LM);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                R vr[VW], vi[VW];
                vload<R, VW>(vr, Xb + k * LN + g * 16 * VW);
                vload<R, VW>(vi, Xb + P + k * LN + g * 16 * VW);
#pragma unroll
                for (int n = 0; n < VW; ++n) { xr[s][g * VW + n] = vr[n]; xi[s][g * VW + n] = vi[n]; }
            }
        }
    }
    // complex product from THREE real products (Gauss): P1 = Re A Re X, P2 = Im A Im X, P3 = (Re A + Im A)(Re X + Im X);
    // Re = P1 - P2, Im = P3 - P1 - P2.  A quarter fewer MFMAs (the f64 matrix pipe of this part sustains ~49 TFLOP/s,
    // scripts/clock_in_kernel.hip, and bounds the multiply) for two extra additions per operand element.
    template <typename T4>
    __device__ inline void mma3(T4 (&p1)[MS][NT], T4 (&p2)[MS][NT], T4 (&p3)[MS][NT]) const {
#pragma unroll
        for (int s = 0; s < KSL; ++s) {
            R sx[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sx[nt] = xr[s][nt] + xi[s][nt];
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                R const sa = ar[s][ms] + ai[s][ms];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    p1[ms][nt] = Acc<R>::mma(ar[s][ms], xr[s][nt], p1[ms][nt]);
                    p2[ms][nt] = Acc<R>::mma(ai[s][ms], xi[s][nt], p2[ms][nt]);
                    p3[ms][nt] = Acc<R>::mma(sa, sx[nt], p3[ms][nt]);
                }
            }
        }
    }
    template <typename T4>
    __device__ inline void mma(T4 (&cre)[MS][NT], T4 (&cim)[MS][NT]) const {
#pragma unroll
        for (int s = 0; s < KSL; ++s)
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                R const nai = -ai[s][ms];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    cre[ms][nt] = Acc<R>::mma(ar[s][ms], xr[s][nt], cre[ms][nt]);
                    cim[ms][nt] = Acc<R>::mma(ar[s][ms], xi[s][nt], cim[ms][nt]);
                    cre[ms][nt] = Acc<R>::mma(nai, xi[s][nt], cre[ms][nt]);
                    cim[ms][nt] = Acc<R>::mma(ai[s][ms], xr[s][nt], cim[ms][nt]);
                }
            }
    }
};

// the vectors an epilogue reads, for the NT neighbouring elements of one lane in one row
template <typename R, int EPI, int NT, bool HASH = false, int STREAMSEL = -1>
struct EpiOps {
    static constexpr bool STREAM = (STREAMSEL < 0) ? (16 * NT * sizeof(R) >= 128) : (STREAMSEL != 0);   // a lane group covers whole 128-byte lines (STREAMSEL: the kernel knows better)
    R ur[NT], ui[NT], xr[NT], xi[NT];
    float wr[NT], wi[NT];
    __device__ inline void load(SpmmArgs const& a, size_t off, int P) {
        if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
            if constexpr (!HASH) { vload_stream<STREAM, float, NT>(wr, a.v3 + off); vload_stream<STREAM, float, NT>(wi, a.v3 + off + P); }   // HASH: recomputed in epilogue_row
        }
        if constexpr (EPI == EPI_XPAY_DOT) if (a.first) {          // first iteration of a solve: old v4 = v8 = 0, not read
#pragma unroll
            for (int n = 0; n < NT; ++n) { ur[n] = 0; ui[n] = 0; xr[n] = 0; xi[n] = 0; }
            return;
        }
        if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
            vload_stream<STREAM, R, NT>(ur, (R const*)a.e0 + off); vload_stream<STREAM, R, NT>(ui, (R const*)a.e0 + off + P);
        }
        if constexpr (EPI == EPI_XPAY_DOT) { vload_stream<STREAM, R, NT>(xr, (R const*)a.e1 + off); vload_stream<STREAM, R, NT>(xi, (R const*)a.e1 + off + P); }
    }
};

// epilogue for VW neighbouring elements at `off` (same arithmetic per element as epilogue<> above); the elements
// are columns n0 .. n0 + VW - 1 of the NT columns of the lane (per-RHS scalars sr/si and partial sums are per column)
template <typename R, int EPI, int VW, int NPL, int NT, bool HASH = false, int LN = 16, typename OPS>
__device__ inline void epilogue_row(SpmmArgs const& a, size_t off, int P, R const (&yr)[VW], R const (&yi)[VW],
                                    R const (&sr)[NT], R const (&si)[NT], int n0, OPS const& o,
                                    uint32_t bq, int eoff, double (&part)[NPL > 0 ? NPL : 1][NT], uint64_t key)
{
    if constexpr (EPI != EPI_RESIDUAL) { vstore_stream<OPS::STREAM, R, VW>((R*)a.Y + off, yr); vstore_stream<OPS::STREAM, R, VW>((R*)a.Y + off + P, yi); }
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
        R nr[VW], ni[VW];
#pragma unroll
        for (int n = 0; n < VW; ++n) {
            R const cr = sr[n0 + n], ci = si[n0 + n];
            if constexpr (EPI == EPI_XPAY_DOT) epi_xpay2(nr[n], ni[n], yr[n], yi[n], o.ur[n], o.ui[n], o.xr[n], o.xi[n], cr, ci);
            else epi_axpy(nr[n], ni[n], yr[n], yi[n], o.ur[n], o.ui[n], cr, ci);
            double wr, wi;   // the shadow vector: read, or recomputed from its hash (tfq_device.hpp)
            if constexpr (HASH) { uint64_t const hq = shadow_quad(key, uint32_t(eoff + n) / (2 * LN), uint32_t(eoff + n) % LN, LN); int const odd = (uint32_t(eoff + n) / LN) & 1; wr = shadow_pick(hq, odd, 0); wi = shadow_pick(hq, odd, 1); }
            else { wr = o.wr[n]; wi = o.wi[n]; }
            epi_dot(part[0][n0 + n], part[1][n0 + n], nr[n], ni[n], wr, wi);
            if constexpr (EPI == EPI_AXPY_NRM_DOT) epi_nrm(part[2][n0 + n], nr[n], ni[n]);
        }
        vstore_stream<OPS::STREAM, R, VW>((R*)a.e0 + off, nr); vstore_stream<OPS::STREAM, R, VW>((R*)a.e0 + off + P, ni);
    } else if constexpr (EPI == EPI_RESIDUAL) {
        R br[VW] = {}, bi[VW] = {};
        if (bq != 0xffffffffu) {
            R const* b = (R const*)a.B + size_t(bq) * 2 * P;
            vload<R, VW>(br, b + eoff); vload<R, VW>(bi, b + eoff + P);
        }
#pragma unroll
        for (int n = 0; n < VW; ++n) {
            R const rr = yr[n] + R(-1) * br[n], ri = yi[n] + R(-1) * bi[n];
            epi_nrm(part[0][n0 + n], rr, ri);
        }
    }
}

// CLAMP: the operand prefetch inside a strip carries no condition (the slice index is clamped to the last slice instead: two
// redundant, cache-resident slice loads per strip), so that the compiler can count the loads in flight and waits for exactly the
// slice it is about to multiply -- with the conditional form it drains the whole queue (s_waitcnt vmcnt(0)) in front of every
// slice.  Used where a strip has many slices (blocks of 32 rows and more: the matrix-pipe-bound shapes).
// NSET: operand register sets of the software pipeline (2, or 4 with CLAMP where a slice is small: the float shapes).
template <typename R, int LM, int LN, int EPI, bool PRE, bool M3, bool HASH, bool CLAMP = false, int NSET = 2>
__global__ __launch_bounds__(256, 2) void k_spmm_mfma(SpmmArgs a) {   // at least 2 waves per SIMD: 256 VGPRs at most
    if (gate_closed(a)) return;
    static_assert(LM % 16 == 0 && LN % 16 == 0, "MFMA tiles are 16 x 16");
    constexpr int P = LM * LN, MT = LM / 16, NT = LN / 16;
    constexpr int MS = RowTiles<R, MT, NT>::MS;      // row tiles per wave
    constexpr int MU = MT / MS;                      // strips per Y block
    constexpr int KSL = (MS * NT >= 4) ? 2 : 4;      // k-steps per slice: bounds the registers of the prefetch
    constexpr int SPP = LM / (4 * KSL);              // slices per block product
    constexpr int NPL = EpiPlanes<EPI>::N;
    constexpr int VW = ColMap<R, NT>::VW, NG = ColMap<R, NT>::NG;
    using T4 = typename Acc<R>::T;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    int const c0 = lc * VW;                          // first block column of this lane, further groups 16 VW apart
    // work groups that are dispatched to the same XCD (blockIdx % 8, observed round-robin) get neighbouring
    // chunks (a.order, tfq_plan.cpp), so that the A blocks shared by neighbouring block columns are served
    // by that XCD's L2
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : a.plainPer ? (blockIdx.x & 7u) * a.plainPer + (blockIdx.x >> 3) : blockIdx.x;
    uint32_t first, last, col = 0;
    if (a.chunkFirst) { first = a.chunkFirst[chunk]; last = a.chunkFirst[chunk + 1]; col = a.chunkCol[chunk]; }
    else { first = min(chunk * a.CH, a.nY); last = min(first + a.CH, a.nY); }

    R sr[NT], si[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { sr[nt] = 0; si[nt] = 0; }
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            sr[nt] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + ColMap<R, NT>::col(lc, nt)];
            si[nt] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + ColMap<R, NT>::col(lc, nt)];
        }
    }
    double part[NPL > 0 ? NPL : 1][NT] = {};

    uint32_t const nUnits = (last - first) * MU;     // unit = strip of MS * 16 rows of one Y block
    using CU32o = __attribute__((address_space(4))) uint32_t const*;
    CU32o const yOrder = (CU32o)(uintptr_t)a.yOrder;
    for (uint32_t u = wave; u < nUnits; u += 4) {
        uint32_t const y = a.yOrder ? yOrder[first + u / MU] : first + u / MU;   // (plain mode with a prepared order: which Y block this position computes)
        int const i0 = int(u % MU) * 16 * MS;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 cre[MS][NT], cim[MS][NT], cp3[M3 ? MS : 1][M3 ? NT : 1];   // M3: P1, P2, P3
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                cre[ms][nt] = T4{0, 0, 0, 0}; cim[ms][nt] = T4{0, 0, 0, 0};
                if constexpr (M3) cp3[ms][nt] = T4{0, 0, 0, 0};
            }
        auto mma = [&](Slice<R, MS, NT, KSL> const& o) __attribute__((always_inline)) {
            if constexpr (M3) o.mma3(cre, cim, cp3); else o.mma(cre, cim);
        };
        // index lists through the constant address space: uniform reads stay scalar loads whatever the stores around them
        using CU32 = __attribute__((address_space(4))) uint32_t const*;
        CU32 const cstarts = (CU32)(uintptr_t)a.starts; CU32 const cpairs = (CU32)(uintptr_t)a.pairs;
        uint32_t const q0 = cstarts[y];
        uint32_t const nT = (cstarts[y + 1] - q0) * SPP;   // slices of this strip
        R const* const A0 = (R const*)a.A + i0 + lc * MS;
        R const* const X0 = (R const*)a.X + c0;
        auto fetch = [&](Slice<R, MS, NT, KSL>& o, uint32_t t) {
            uint32_t const q = q0 + t / SPP;
            int const k0 = int(t % SPP) * (4 * KSL);
            o.template load<LM, LN>(A0 + size_t(cpairs[2 * size_t(q)]) * 2 * LM * LM,
                                    X0 + size_t(cpairs[2 * size_t(q) + 1]) * 2 * P, k0, lr);
        };
        // block row of accumulator register r of row tile ms
        auto row_of = [&](int ms, int r) { return i0 + Acc<R>::row(lane, r) * MS + ms; };
        // software pipeline, two register sets: the loads of slices t+1, t+2 are in flight while the MFMAs
        // of slice t issue.  With PRE the operands of the epilogue (old v4|v5, v8, v3) are requested right
        // behind the first two slices: vmcnt retires in order, so they must be younger than the slices
        // whose MFMAs should start first and they have two slices of matrix work to arrive.
        static_assert(NSET == 2 || CLAMP, "the deeper pipeline is written for the unconditional prefetch");
        Slice<R, MS, NT, KSL> o[NSET];
        if constexpr (CLAMP) {
            if (nT > 0) {
#pragma unroll
                for (int i = 0; i < NSET; ++i) fetch(o[i], (uint32_t(i) < nT) ? uint32_t(i) : nT - 1);
            }
        } else {
            if (nT > 0) fetch(o[0], 0);
            if (nT > 1) fetch(o[1], 1);
        }
        EpiOps<R, EPI, VW, HASH> ops[PRE ? MS * 4 * NG : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        ops[(ms * 4 + r) * NG + g].load(a, size_t(y) * 2 * P + row_of(ms, r) * LN + c0 + g * 16 * VW, P);
        }
        uint32_t t = 0;
        for (; t + NSET <= nT; t += NSET) {
#pragma unroll
            for (int i = 0; i < NSET; ++i) {
                mma(o[i]);
                uint32_t const tn = t + NSET + i;
                if constexpr (CLAMP) {
                    // (pinning the loads right behind the MFMAs of their set with sched_barrier was measured and is slower:
                    //  32 x 32 c with cache-hot operands 102.6 -> 95.0 TFLOP/s; hipcc's own interleaving is kept)
                    fetch(o[i], (tn < nT) ? tn : nT - 1);
                } else if (tn < nT) fetch(o[i], tn);
            }
        }
#pragma unroll
        for (int i = 0; i < NSET - 1; ++i) if (t + i < nT) mma(o[i]);

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < NG; ++g) {       // accesses of at most 16 bytes per lane
                    int const e = row_of(ms, r) * LN + c0 + g * 16 * VW;
                    size_t const off = size_t(y) * 2 * P + e;
                    R yr[VW], yi[VW];
#pragma unroll
                    for (int n = 0; n < VW; ++n) {
                        if constexpr (M3) {
                            R const p1 = cre[ms][g * VW + n][r], p2 = cim[ms][g * VW + n][r];
                            yr[n] = p1 - p2; yi[n] = (cp3[ms][g * VW + n][r] - p1) - p2;
                        } else { yr[n] = cre[ms][g * VW + n][r]; yi[n] = cim[ms][g * VW + n][r]; }
                    }
                    if constexpr (!PRE) ops[0].load(a, off, P);
                    epilogue_row<R, EPI, VW, NPL, NT, HASH, LN>(a, off, P, yr, yi, sr, si, g * VW, ops[PRE ? (ms * 4 + r) * NG + g : 0], bq, e, part, key);
                }
    }

    if constexpr (NPL > 0) {
        // rows live on lane/16 (and registers): add the four lane groups, then the four waves in order
        __shared__ double s[4][NPL][LN];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double v = part[p][nt];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 16) s[wave][p][ColMap<R, NT>::col(lane, nt)] = v;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, j = e % LN;
            double const sum = ((s[0][p][j] + s[1][p][j]) + s[2][p][j]) + s[3][p][j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 16 x 16 complex<double> on the ROW-PAIR-INTERLEAVED element order (tfq_device.hpp: plane[r/2][s][r%2] for every block of
// the plan, A blocks with r = k): the structure of k_spmm_mfma (one wave owns the 16 x 16 strip of a Y block in MFMA
// accumulators, two operand register sets, epilogue operands requested behind the first two block products), but every
// access is 16 bytes per lane -- one wave instruction moves 1 KiB instead of 512 bytes:
//   k-steps: lane group lr = lane / 16 loads the k pairs lr and lr + 4, i.e. k = 2 lr, 2 lr + 1, 2 lr + 8, 2 lr + 9 feed the
//            four MFMA steps of a block product (which k a step contracts is free as long as A and X agree);
//   rows:    lane column a supplies A row rowp(a) = 2 (a % 4 + 4 (a / 8)) + (a / 4) % 2, so that the accumulator registers
//            (0, 1) and (2, 3) of lane group lr are the row pairs (2 lr, 2 lr + 1), (2 lr + 8, 2 lr + 9) of column lane % 16:
//            the epilogue reads and writes them as two 16-byte accesses per vector and plane.
// Measured on P2 against k_spmm_mfma on the native order (same box, scripts/lab, profiles/r02_lab.txt): fused multiplies
// 0.684 / 0.660 -> 0.628 / 0.595 ms.  The sums of a block product run over k in another order than in the native kernel
// (results differ in the last bits, within the tolerances of the parity tests).
__device__ inline int ilv_rowp(int a) { return 2 * ((a & 3) + 4 * (a >> 3)) + ((a >> 2) & 1); }
using d2v = __attribute__((ext_vector_type(2))) double;
using f2v = __attribute__((ext_vector_type(2))) float;

// ANT: the A operands are loaded non-temporally.  For an operator applied to one or two block columns every A block is used
// once per multiply -- the kernel is a stream of A through HBM, and 16-byte non-temporal loads take it from 5.5 to 6.6 TB/s
// (one block column, 1.3 GB of A: plain multiply 0.69 -> 0.82 of 8 TB/s, fused 0.76 -> 0.85, profiles/r02_lab.txt); with
// many columns A is re-used out of the caches and must stay there (the plan decides: SpmmArgs::aOnce).
template <int EPI, bool HASH, bool ANT = false, bool FIRST = false>   // FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256, 2) void k_spmm_ilv16(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    // the index lists through the constant address space: uniform reads become scalar loads whatever the stores around them
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;   // XCD-aware launch order (tfq_plan.cpp)
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    R sr = 0, si = 0;
    if constexpr (UPD) { sr = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc]; si = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc]; }
    double part[NPL > 0 ? NPL : 1] = {};
    __shared__ double s[4][NPL > 0 ? NPL : 1][LN];

    struct Ops { d2v ar[2], ai[2], xr[2], xi[2]; };   // [k pair lr | lr + 4]
    R const* const A0 = (R const*)a.A + (lr * 16 + ilv_rowp(lc)) * 2;
    R const* const X0 = (R const*)a.X + (lr * 16 + lc) * 2;
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * P;
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if constexpr (ANT) { o.ar[h] = __builtin_nontemporal_load((d2v const*)(Ab + h * 128)); o.ai[h] = __builtin_nontemporal_load((d2v const*)(Ab + P + h * 128)); }
            else { o.ar[h] = *(d2v const*)(Ab + h * 128); o.ai[h] = *(d2v const*)(Ab + P + h * 128); }
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
        // the epilogue operands of EPI_AXPY_NRM_DOT (4 loads) are requested in FRONT of the first two block products' operands, those of
        // EPI_XPAY_DOT (8 loads) behind them: measured both ways, profiles/r02_ab_traversal.txt (vmcnt retires in order)
        constexpr bool EPI_FIRST = (EPI == EPI_AXPY_NRM_DOT);
        if constexpr (!EPI_FIRST) {
            if (q0 < q1) fetch(o0, q0);
            if (q0 + 1 < q1) fetch(o1, q0 + 1);
        }
        // this lane's elements of the Y block: rows (2 lr, 2 lr + 1) and (2 lr + 8, 2 lr + 9) of column lc
        int const eb[2] = { (lr * 16 + lc) * 2, ((lr + 4) * 16 + lc) * 2 };
        size_t const yoff = size_t(y) * 2 * P;
        d2v ur[2], ui[2], vr[2], vi[2]; f2v wr[2], wi[2];
        if constexpr (UPD) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {   // old v4 | v5, v8, v3: touched once, non-temporal
                if constexpr (EPI == EPI_XPAY_DOT && FIRST) { ur[h] = d2v{0, 0}; ui[h] = d2v{0, 0}; vr[h] = d2v{0, 0}; vi[h] = d2v{0, 0}; }   // first iteration: old v4 = v8 = 0, not read
                else {
                ur[h] = ld_stream<true>((d2v const*)((R const*)a.e0 + yoff + eb[h])); ui[h] = ld_stream<true>((d2v const*)((R const*)a.e0 + yoff + eb[h] + P));
                if constexpr (EPI == EPI_XPAY_DOT) { vr[h] = ld_stream<true>((d2v const*)((R const*)a.e1 + yoff + eb[h])); vi[h] = ld_stream<true>((d2v const*)((R const*)a.e1 + yoff + eb[h] + P)); }
                }
                if constexpr (!HASH) { wr[h] = __builtin_nontemporal_load((f2v const*)(a.v3 + yoff + eb[h])); wi[h] = __builtin_nontemporal_load((f2v const*)(a.v3 + yoff + eb[h] + P)); }
            }
        }
        if constexpr (EPI_FIRST) {
            if (q0 < q1) fetch(o0, q0);
            if (q0 + 1 < q1) fetch(o1, q0 + 1);
        }
        // (r03, profiles/r03_ab_exact_waits.txt: the conditional prefetches make the compiler wait with vmcnt(0) in front of every pair of
        //  products; both forms with exact waits -- prefetch index clamped to the last product, or straight-line tails behind a loop that
        //  always prefetches -- measured 4-8 % SLOWER on P2: redundant cache-hot fetches, or 192 VGPRs = two waves per SIMD)
        uint32_t q = q0;
        for (; q + 2 <= q1; q += 2) {
            mma(o0);
            if (q + 2 < q1) fetch(o0, q + 2);
            mma(o1);
            if (q + 3 < q1) fetch(o1, q + 3);
        }
        if (q < q1) mma(o0);

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // the shadow vector recomputed: one hash for this pair of rows (tfq_device.hpp: shadow_quad).  Drawn here, inside the loop:
            // both hashes in front of it cost spmm_v4_dot 2 % (0.626 against 0.614 ms on P2, profiles/r02_ab_hash.txt)
            uint64_t const hqh = HASH ? shadow_quad(key, uint32_t(lr + 4 * h), uint32_t(lc), LN) : 0;
            d2v yr, yi, nr, ni;
            d2v br = d2v{0, 0}, bi = d2v{0, 0};
            if constexpr (EPI == EPI_RESIDUAL) if (bq != 0xffffffffu) {
                R const* b = (R const*)a.B + size_t(bq) * 2 * P;
                br = *(d2v const*)(b + eb[h]); bi = *(d2v const*)(b + eb[h] + P);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                yr[e] = cre[2 * h + e]; yi[e] = cim[2 * h + e];
                // explicit fused multiply-adds: the HASH and the v3-reading instance of this kernel must round alike
                // (tests/test_gpu_hash_mode.py compares them bit by bit), whatever the compiler would contract on its own
                if constexpr (EPI == EPI_XPAY_DOT) {         // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                    R const tr = __builtin_fma(-si, ui[h][e], __builtin_fma(sr, ur[h][e], vr[h][e]));
                    R const ti = __builtin_fma(sr, ui[h][e], __builtin_fma(si, ur[h][e], vi[h][e]));
                    nr[e] = __builtin_fma(-si, ti, __builtin_fma(sr, tr, yr[e]));
                    ni[e] = __builtin_fma(sr, ti, __builtin_fma(si, tr, yi[e]));
                } else if constexpr (EPI == EPI_AXPY_NRM_DOT) { // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
                    nr[e] = __builtin_fma(-si, yi[e], __builtin_fma(sr, yr[e], ur[h][e]));
                    ni[e] = __builtin_fma(sr, yi[e], __builtin_fma(si, yr[e], ui[h][e]));
                }
                if constexpr (UPD) {
                    double w0, w1;      // the logical elements (rows 2 (lr + 4 h) + e, column lc) are one quad of the shadow vector's hash
                    if constexpr (HASH) { w0 = shadow_pick(hqh, e, 0); w1 = shadow_pick(hqh, e, 1); }
                    else { w0 = wr[h][e]; w1 = wi[h][e]; }
                    double const dr = nr[e], di = ni[e];
                    part[0] = __builtin_fma(-di, w1, __builtin_fma(dr, w0, part[0]));
                    part[1] = __builtin_fma(di, w0, __builtin_fma(dr, w1, part[1]));
                    if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[2]));
                } else if constexpr (EPI == EPI_RESIDUAL) {     // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
                    R const rr = yr[e] + R(-1) * br[e], ri = yi[e] + R(-1) * bi[e];
                    double const dr = rr, di = ri;
                    part[0] += dr * dr + di * di;
                }
            }
            if constexpr (EPI != EPI_RESIDUAL) {
                st_stream<true>((d2v*)((R*)a.Y + yoff + eb[h]), yr); st_stream<true>((d2v*)((R*)a.Y + yoff + eb[h] + P), yi);
            }
            if constexpr (UPD) {
                st_stream<true>((d2v*)((R*)a.e0 + yoff + eb[h]), nr); st_stream<true>((d2v*)((R*)a.e0 + yoff + eb[h] + P), ni);
            }
        }
    }
    if constexpr (NPL > 0) {
        // rows live on lane / 16 (and registers): add the four lane groups, then the four waves in order
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
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 16 x 16 complex<float> with groups of FOUR rows interleaved (plane[r/4][s][r%4]): the float counterpart of k_spmm_ilv16.
// A lane of v_mfma_f32_16x16x4_f32 loads the k quad lr = lane / 16 (k = 4 lr .. 4 lr + 3) of its column as ONE 16-byte access --
// MFMA step e contracts k = 4 lr + e -- and its four accumulator registers are the rows 4 lr .. 4 lr + 3 of column lane % 16
// (the C layout of the f32 instruction), i.e. again one 16-byte piece of every epilogue vector: a block product takes 4 wave-wide
// loads of 1 KiB (16 of 256 bytes in k_spmm_mfma<float, 16, 16>), an epilogue 2 accesses per vector.  16 x 16 in float is the
// default shape of the reference's own benchmark (`bench_tfqmrgpu multi`, bench_tfqmrgpu.cu:445-450).
using f4v = __attribute__((ext_vector_type(4))) float;

template <int EPI, bool HASH, bool ANT = false, bool FIRST = false>   // FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256, 2) void k_spmm_ilv16f(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = float;
    constexpr int LN = 16, P = 256, NPL = EpiPlanes<EPI>::N;
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    R sr = 0, si = 0;
    if constexpr (UPD) { sr = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + lc]; si = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + lc]; }
    double part[NPL > 0 ? NPL : 1] = {};
    __shared__ double s[4][NPL > 0 ? NPL : 1][LN];

    int const mine = (lr * 16 + lc) * 4;                       // this lane's 16 bytes of a plane: quad lr, column (or A row) lc
    struct Ops { f4v ar, ai, xr, xi; };
    R const* const A0 = (R const*)a.A + mine;
    R const* const X0 = (R const*)a.X + mine;
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * P;
        R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
        if constexpr (ANT) { o.ar = __builtin_nontemporal_load((f4v const*)Ab); o.ai = __builtin_nontemporal_load((f4v const*)(Ab + P)); }
        else { o.ar = *(f4v const*)Ab; o.ai = *(f4v const*)(Ab + P); }
        o.xr = *(f4v const*)Xb; o.xi = *(f4v const*)(Xb + P);
    };
    for (uint32_t u = wave; u < last - first; u += 4) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        f4 cre = f4{0, 0, 0, 0}, cim = f4{0, 0, 0, 0};
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                R const nai = -o.ai[e];
                cre = Acc<R>::mma(o.ar[e], o.xr[e], cre);
                cim = Acc<R>::mma(o.ar[e], o.xi[e], cim);
                cre = Acc<R>::mma(nai, o.xi[e], cre);
                cim = Acc<R>::mma(o.ai[e], o.xr[e], cim);
            }
        };
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        Ops o0, o1;
        constexpr bool EPI_FIRST = true;   // epilogue operands requested in front of the first products' operands: -1 % (profiles/r02_ab_traversal.txt)
        if constexpr (!EPI_FIRST) {
            if (q0 < q1) fetch(o0, q0);
            if (q0 + 1 < q1) fetch(o1, q0 + 1);
        }
        size_t const yoff = size_t(y) * 2 * P + mine;          // rows 4 lr .. 4 lr + 3 of column lc
        f4v ur, ui, vr, vi, wr, wi;
        if constexpr (UPD) {
            if constexpr (EPI == EPI_XPAY_DOT && FIRST) { ur = f4v{0, 0, 0, 0}; ui = ur; vr = ur; vi = ur; }   // first iteration: old v4 = v8 = 0, not read
            else {
            ur = __builtin_nontemporal_load((f4v const*)((R const*)a.e0 + yoff)); ui = __builtin_nontemporal_load((f4v const*)((R const*)a.e0 + yoff + P));
            if constexpr (EPI == EPI_XPAY_DOT) { vr = __builtin_nontemporal_load((f4v const*)((R const*)a.e1 + yoff)); vi = __builtin_nontemporal_load((f4v const*)((R const*)a.e1 + yoff + P)); }
            }
            if constexpr (!HASH) { wr = __builtin_nontemporal_load((f4v const*)(a.v3 + yoff)); wi = __builtin_nontemporal_load((f4v const*)(a.v3 + yoff + P)); }
        }
        if constexpr (EPI_FIRST) {
            if (q0 < q1) fetch(o0, q0);
            if (q0 + 1 < q1) fetch(o1, q0 + 1);
        }
        uint32_t q = q0;
        for (; q + 2 <= q1; q += 2) {
            mma(o0);
            if (q + 2 < q1) fetch(o0, q + 2);
            mma(o1);
            if (q + 3 < q1) fetch(o1, q + 3);
        }
        if (q < q1) mma(o0);

        uint64_t hq[2] = {0, 0};   // the shadow vector recomputed: one hash per pair of rows (tfq_device.hpp: shadow_quad)
        if constexpr (HASH) { hq[0] = shadow_quad(key, uint32_t(2 * lr), uint32_t(lc), LN); hq[1] = shadow_quad(key, uint32_t(2 * lr + 1), uint32_t(lc), LN); }
        f4v yr, yi, nr, ni;
        f4v br = f4v{0, 0, 0, 0}, bi = f4v{0, 0, 0, 0};
        if constexpr (EPI == EPI_RESIDUAL) {
            uint32_t const bq = a.bOfX ? a.bOfX[y] : y;
            if (bq != 0xffffffffu) { R const* b = (R const*)a.B + size_t(bq) * 2 * P + mine; br = *(f4v const*)b; bi = *(f4v const*)(b + P); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            yr[e] = cre[e]; yi[e] = cim[e];
            // explicit fused multiply-adds: the HASH and the v3-reading instance must round alike (tests compare them bit by bit)
            if constexpr (EPI == EPI_XPAY_DOT) {         // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                R const tr = __builtin_fmaf(-si, ui[e], __builtin_fmaf(sr, ur[e], vr[e]));
                R const ti = __builtin_fmaf(sr, ui[e], __builtin_fmaf(si, ur[e], vi[e]));
                nr[e] = __builtin_fmaf(-si, ti, __builtin_fmaf(sr, tr, yr[e]));
                ni[e] = __builtin_fmaf(sr, ti, __builtin_fmaf(si, tr, yi[e]));
            } else if constexpr (EPI == EPI_AXPY_NRM_DOT) { // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
                nr[e] = __builtin_fmaf(-si, yi[e], __builtin_fmaf(sr, yr[e], ur[e]));
                ni[e] = __builtin_fmaf(sr, yi[e], __builtin_fmaf(si, yr[e], ui[e]));
            }
            if constexpr (UPD) {
                double w0, w1;          // rows 4 lr + e of column lc: two quads of the shadow vector's hash
                if constexpr (HASH) { w0 = shadow_pick(hq[e >> 1], e & 1, 0); w1 = shadow_pick(hq[e >> 1], e & 1, 1); }
                else { w0 = wr[e]; w1 = wi[e]; }
                double const dr = nr[e], di = ni[e];
                part[0] = __builtin_fma(-di, w1, __builtin_fma(dr, w0, part[0]));
                part[1] = __builtin_fma(di, w0, __builtin_fma(dr, w1, part[1]));
                if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[2]));
            } else if constexpr (EPI == EPI_RESIDUAL) {     // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
                R const rr = yr[e] + R(-1) * br[e], ri = yi[e] + R(-1) * bi[e];
                double const dr = rr, di = ri;
                part[0] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[0]));
            }
        }
        if constexpr (EPI != EPI_RESIDUAL) { __builtin_nontemporal_store(yr, (f4v*)((R*)a.Y + yoff)); __builtin_nontemporal_store(yi, (f4v*)((R*)a.Y + yoff + P)); }
        if constexpr (UPD) { __builtin_nontemporal_store(nr, (f4v*)((R*)a.e0 + yoff)); __builtin_nontemporal_store(ni, (f4v*)((R*)a.e0 + yoff + P)); }
    }
    if constexpr (NPL > 0) {
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
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// complex<float> blocks of 16 | 32 rows and 32 columns on the quad-interleaved element order (32 x 32 = BASELINE config 3; written for
// LM, LN multiples of 16; with 64 columns and all four column tiles in one wave it was level with (16 x 64, 32 x 64) or 10 % behind (64 x 64)
// k_spmm_mfma in round 2 -- four tiles of accumulators and operands cost a wave per SIMD; since round 3 a wave takes half of the columns, NH below):
// k_spmm_ilv16f's access pattern with MS x NT MFMA tiles per wave.  A wave owns a strip of MS * 16 rows of a Y block (MS = 2 where the
// block has two row tiles and 32 columns, else 1: the accumulators stay within 32 VGPRs).  A slice is one group of four k quads (quads
// lr + 4 m, 16 k values): MS + NT pairs of wave-wide 1-KiB loads feed 16 MS NT MFMAs (32 x 32: 8 loads for 64 MFMAs, against 16 loads of 512
// bytes in k_spmm_mfma<float, 32, 32>), and every accumulator tile is one 16-byte piece of each epilogue vector (8-byte pieces there).
// The ablations of profiles/r02_ab_config3.txt are why: that kernel gains time with every operand load instruction that is removed.
// No epilogue-operand prefetch (the registers of the tiles: three waves per SIMD matter more), v3 is read.
template <int LM, int LN, int EPI, bool FIRST = false>   // FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256, 3) void k_spmm_ilvf(SpmmArgs a) {   // two column tiles per wave: three waves per SIMD (168 VGPRs at most)
    if (gate_closed(a)) return;
    using R = float;
    constexpr int P = LM * LN, Q = LM * LM, MT = LM / 16, NPL = EpiPlanes<EPI>::N;
    // 64 columns (r03): a wave works on ONE half of the columns (NH = 2 halves of two tiles; the units of a Y block are dealt (strip, half) with the
    // half running fastest, so wave w of a work group keeps half w % 2 and its per-column scalars and sums) -- all four tiles in one wave need the
    // registers of two waves per SIMD, and the epilogue stream then does not overlap with the matrix work (32 x 64: 0.526 ms = multiply 0.354 + stream)
    constexpr int NH = (LN / 16 > 2) ? LN / 32 : 1, NT = LN / 16 / NH;
    constexpr int MS = (MT % 2 == 0 && NT <= 2) ? 2 : 1;          // row tiles per wave
    constexpr int MU = MT / MS;                                   // strips per Y block
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    int const c0 = (NH > 1) ? (wave % NH) * 16 * NT : 0;      // first column of this wave's half
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    R sr[NT], si[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { sr[nt] = 0; si[nt] = 0; }
    if constexpr (UPD) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            sr[nt] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + c0 + lc + 16 * nt];
            si[nt] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + c0 + lc + 16 * nt];
        }
    }
    double part[NPL > 0 ? NPL : 1][NT] = {};
    __shared__ double s[4][NPL > 0 ? NPL : 1][LN];

    // 16 bytes of a plane: quad g (rows | k values 4 g .. 4 g + 3) of column c of an X-shaped block, of row c of a (transposed) A block
    auto pieceX = [](int g, int c) { return (g * LN + c) * 4; };
    auto pieceA = [](int g, int c) { return (g * LM + c) * 4; };
    struct Ops { f4v ar[MS], ai[MS], xr[NT], xi[NT]; };
    uint32_t const nUnits = (last - first) * MU * NH;             // unit = strip of MS * 16 rows of one Y block [x half of its columns]
    for (uint32_t u = wave; u < nUnits; u += 4) {                 // (u % NH == wave % NH: 4 is a multiple of NH)
        uint32_t const y = first + (u / NH) / MU;
        int const t0 = int((u / NH) % MU) * MS;                   // first row tile of the strip
        auto fetch = [&](Ops& o, uint32_t q, int m) __attribute__((always_inline)) {
            R const* Ab = (R const*)a.A + size_t(pairs[2 * size_t(q)]) * 2 * Q;
            R const* Xb = (R const*)a.X + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
            // tile by tile, A and X in turn: vmcnt retires in order and the first MFMAs need the first tiles of both (3 % on 32 x 32)
#pragma unroll
            for (int t = 0; t < (MS > NT ? MS : NT); ++t) {
                if (t < MS) {
                    int const at = pieceA(lr + 4 * m, lc + 16 * (t0 + t));
                    o.ar[t] = *(f4v const*)(Ab + at); o.ai[t] = *(f4v const*)(Ab + Q + at);
                }
                if (t < NT) {
                    int const at = pieceX(lr + 4 * m, c0 + lc + 16 * t);
                    o.xr[t] = *(f4v const*)(Xb + at); o.xi[t] = *(f4v const*)(Xb + P + at);
                }
            }
        };
        f4 cre[MS][NT], cim[MS][NT];
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { cre[ms][nt] = f4{0, 0, 0, 0}; cim[ms][nt] = f4{0, 0, 0, 0}; }
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 4; ++e)                           // MFMA step e contracts k = 4 (lr + 4 m) + e
#pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    R const nai = -o.ai[ms][e];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        cre[ms][nt] = Acc<R>::mma(o.ar[ms][e], o.xr[nt][e], cre[ms][nt]);
                        cim[ms][nt] = Acc<R>::mma(o.ar[ms][e], o.xi[nt][e], cim[ms][nt]);
                        cre[ms][nt] = Acc<R>::mma(nai, o.xi[nt][e], cre[ms][nt]);
                        cim[ms][nt] = Acc<R>::mma(o.ai[ms][e], o.xr[nt][e], cim[ms][nt]);
                    }
                }
        };
        // the slices of the strip in one sequence: slice t = k group (t % MT) of block product q0 + t / MT; two register sets
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        Ops o0, o1;
        if constexpr (2 == MT) {          // the two register sets are the two slices of a block product (2-5 % faster than the general form below)
            if (q0 < q1) { fetch(o0, q0, 0); fetch(o1, q0, 1); }
            for (uint32_t q = q0; q < q1; ++q) {
                mma(o0);
                if (q + 1 < q1) fetch(o0, q + 1, 0);
                mma(o1);
                if (q + 1 < q1) fetch(o1, q + 1, 1);
            }
        } else {
            uint32_t const nT = (q1 - q0) * MT;
            if (nT > 0) fetch(o0, q0, 0);
            if (nT > 1) fetch(o1, q0 + 1 / MT, 1 % MT);          // (MT == 1: slice 1 is the next block product)
            for (uint32_t t = 0; t < nT; t += 2) {
                mma(o0);
                if (t + 2 < nT) fetch(o0, q0 + (t + 2) / MT, int((t + 2) % MT));
                if (t + 1 < nT) mma(o1);
                if (t + 3 < nT) fetch(o1, q0 + (t + 3) / MT, int((t + 3) % MT));
            }
        }

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
#pragma unroll
        for (int ms = 0; ms < MS; ++ms)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                // accumulator registers 0 .. 3 of tile (ms, nt): rows 16 (t0 + ms) + 4 lr .. + 3 of column 16 nt + lc = one 16-byte piece
                int const at = pieceX(4 * (t0 + ms) + lr, c0 + 16 * nt + lc);
                size_t const yoff = size_t(y) * 2 * P + at;
                f4v ur, ui, vr, vi, wr, wi;
                if constexpr (UPD) {
                    if constexpr (EPI == EPI_XPAY_DOT && FIRST) { ur = f4v{0, 0, 0, 0}; ui = ur; vr = ur; vi = ur; }   // first iteration: old v4 = v8 = 0, not read
                    else {
                    ur = __builtin_nontemporal_load((f4v const*)((R const*)a.e0 + yoff)); ui = __builtin_nontemporal_load((f4v const*)((R const*)a.e0 + yoff + P));
                    if constexpr (EPI == EPI_XPAY_DOT) { vr = __builtin_nontemporal_load((f4v const*)((R const*)a.e1 + yoff)); vi = __builtin_nontemporal_load((f4v const*)((R const*)a.e1 + yoff + P)); }
                    }
                    wr = __builtin_nontemporal_load((f4v const*)(a.v3 + yoff)); wi = __builtin_nontemporal_load((f4v const*)(a.v3 + yoff + P));
                }
                f4v br = f4v{0, 0, 0, 0}, bi = f4v{0, 0, 0, 0};
                if constexpr (EPI == EPI_RESIDUAL) if (bq != 0xffffffffu) {
                    R const* b = (R const*)a.B + size_t(bq) * 2 * P + at;
                    br = *(f4v const*)b; bi = *(f4v const*)(b + P);
                }
                f4v yr, yi, nr, ni;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    yr[e] = cre[ms][nt][e]; yi[e] = cim[ms][nt][e];
                    if constexpr (EPI == EPI_XPAY_DOT) {         // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                        R const tr = __builtin_fmaf(-si[nt], ui[e], __builtin_fmaf(sr[nt], ur[e], vr[e]));
                        R const ti = __builtin_fmaf(sr[nt], ui[e], __builtin_fmaf(si[nt], ur[e], vi[e]));
                        nr[e] = __builtin_fmaf(-si[nt], ti, __builtin_fmaf(sr[nt], tr, yr[e]));
                        ni[e] = __builtin_fmaf(sr[nt], ti, __builtin_fmaf(si[nt], tr, yi[e]));
                    } else if constexpr (EPI == EPI_AXPY_NRM_DOT) { // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
                        nr[e] = __builtin_fmaf(-si[nt], yi[e], __builtin_fmaf(sr[nt], yr[e], ur[e]));
                        ni[e] = __builtin_fmaf(sr[nt], yi[e], __builtin_fmaf(si[nt], yr[e], ui[e]));
                    }
                    if constexpr (UPD) {
                        double const w0 = wr[e], w1 = wi[e], dr = nr[e], di = ni[e];
                        part[0][nt] = __builtin_fma(-di, w1, __builtin_fma(dr, w0, part[0][nt]));
                        part[1][nt] = __builtin_fma(di, w0, __builtin_fma(dr, w1, part[1][nt]));
                        if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2][nt] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[2][nt]));
                    } else if constexpr (EPI == EPI_RESIDUAL) {     // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
                        R const rr = yr[e] + R(-1) * br[e], ri = yi[e] + R(-1) * bi[e];
                        double const dr = rr, di = ri;
                        part[0][nt] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[0][nt]));
                    }
                }
                if constexpr (EPI != EPI_RESIDUAL) { __builtin_nontemporal_store(yr, (f4v*)((R*)a.Y + yoff)); __builtin_nontemporal_store(yi, (f4v*)((R*)a.Y + yoff + P)); }
                if constexpr (UPD) { __builtin_nontemporal_store(nr, (f4v*)((R*)a.e0 + yoff)); __builtin_nontemporal_store(ni, (f4v*)((R*)a.e0 + yoff + P)); }
            }
    }
    if constexpr (NPL > 0) {
        // rows live on lane / 16 (and registers): add the four lane groups, then the four waves in order
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double v = part[p][nt];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 16) s[wave][p][c0 + lane + 16 * nt] = v;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, j = e % LN;
            double sum;
            if constexpr (NH > 1) sum = s[(j / (16 * NT)) % NH][p][j] + s[(j / (16 * NT)) % NH + 2][p][j];   // the two waves of this column's half
            else sum = ((s[0][p][j] + s[1][p][j]) + s[2][p][j]) + s[3][p][j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// (r03: k_spmm_ilvz, the wide complex<double> shapes on the row-pair-interleaved order with one column tile per wave and three waves per SIMD, was
//  4-22 % slower than k_spmm_mfma below -- twice the operand loads per MFMA; two tiles per wave spill at 168 VGPRs -- and is in the git history only:
//  commit 167d902, profiles/r03_ab_ilvz.txt)

// ---------------------------------------------------------------------------------------------------
// 8 x 8 complex<double> on the row-pair-interleaved element order (BASELINE config 5: the bandwidth-bound shape).
// A block is 1 KiB = ONE wave-wide 16-byte access: lane (lr = lane / 16, c = (lane % 16) / 8, j = lane % 8) holds the k pair lr
// (k = 2 lr, 2 lr + 1) of plane c (Re | Im) and column j.  The matrix tile is filled like in k_spmm_mfma8:
//      [Re A; Im A] (16 x 8)  x  [Re X | Im X] (8 x 16)  =  [Q00 Q01; Q10 Q11],   Y = (Q00 - Q11) + i (Q01 + Q10)
// with the A rows supplied in the order pi(a) = 2 (a % 4) + a / 4, so that accumulator registers (0, 1) | (2, 3) of a lane are the
// rows (2 lr, 2 lr + 1) of Q0c | Q1c: one exchange with the lane 8 further (the other plane) gives every lane one 16-byte piece
// of Re Y (c = 0) or Im Y (c = 1) -- the Y block, each epilogue operand and each result are again ONE access per wave.
// Per Y block: 2 loads per block product + 2 (3) epilogue loads + 2 stores, against 4 per product + 8 + 4 eight-byte accesses and
// an LDS round trip in k_spmm_mfma8.
__device__ inline double xor8(double v) { return __shfl_xor(v, 8); }
__device__ inline d2v xor8(d2v v) { return d2v{__shfl_xor(v[0], 8), __shfl_xor(v[1], 8)}; }

template <int EPI, bool HASH, bool FIRST = false>   // FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256) void k_spmm_ilv8(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LN = 8, P = 64, NPL = EpiPlanes<EPI>::N;
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15, cp = lc >> 3, j = lc & 7;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    R sr = 0, si = 0;
    if constexpr (UPD) { sr = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + j]; si = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + j]; }
    double part[NPL > 0 ? NPL : 1] = {};
    __shared__ double s[4][NPL > 0 ? NPL : 1][LN];

    int const mine = cp * P + (lr * 8 + j) * 2;                                 // this lane's 16 bytes of an X-shaped block
    R const* const A0 = (R const*)a.A + cp * P + (lr * 8 + 2 * (j & 3) + (j >> 2)) * 2;   // A: row pi(j) of plane cp, k pair lr
    R const* const X0 = (R const*)a.X + mine;
    struct Ops { d2v av, xv; };
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        o.av = *(d2v const*)(A0 + size_t(pairs[2 * size_t(q)]) * 2 * P);
        o.xv = *(d2v const*)(X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P);
    };
    for (uint32_t u = wave; u < last - first; u += 4) {
        uint32_t const y = first + u;
        uint64_t const key = HASH ? shadow_key(uint32_t(a.origCol[col]), a.rowI[y]) : 0;
        T4 acc = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], nq = starts[y + 1] - q0;
        constexpr int DEPTH = 4;
        Ops o[DEPTH];
        constexpr bool EPI_FIRST = true;   // epilogue operands requested in front of the first products' operands: -1 % (profiles/r02_ab_traversal.txt)
        if constexpr (!EPI_FIRST) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) if (uint32_t(dd) < nq) fetch(o[dd], q0 + dd);
        }
        size_t const yoff = size_t(y) * 2 * P + mine;
        d2v uM = d2v{0, 0}, vM = d2v{0, 0}; f2v wM = f2v{0, 0};
        if constexpr (UPD) {                       // the epilogue operands travel while the products are computed
            if constexpr (!(EPI == EPI_XPAY_DOT && FIRST)) {   // (first iteration of a solve: old v4 = v8 = 0, not read)
            uM = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + yoff));
            if constexpr (EPI == EPI_XPAY_DOT) vM = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + yoff));
            }
            if constexpr (!HASH) wM = __builtin_nontemporal_load((f2v const*)(a.v3 + yoff));
        }
        if constexpr (EPI_FIRST) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) if (uint32_t(dd) < nq) fetch(o[dd], q0 + dd);
        }
        for (uint32_t base = 0; base < nq; base += DEPTH) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) {
                if (base + dd < nq) {
                    acc = Acc<R>::mma(o[dd].av[0], o[dd].xv[0], acc);
                    acc = Acc<R>::mma(o[dd].av[1], o[dd].xv[1], acc);
                    if (base + dd + DEPTH < nq) fetch(o[dd], q0 + base + dd + DEPTH);
                }
            }
        }
        // lanes of plane 0 hold (Q00, Q10), lanes of plane 1 (Q01, Q11), rows 2 lr and 2 lr + 1: Re Y = Q00 - Q11, Im Y = Q01 + Q10
        d2v const qa = d2v{acc[0], acc[1]}, qb = xor8(d2v{acc[2], acc[3]});
        d2v const yM = cp ? d2v{qa[0] + qb[0], qa[1] + qb[1]} : d2v{qa[0] - qb[0], qa[1] - qb[1]};   // this lane's plane of Y
        d2v const yO = xor8(yM);
        d2v const yr = cp ? yO : yM, yi = cp ? yM : yO;
        if constexpr (UPD) {
            d2v const uO = xor8(uM), ur = cp ? uO : uM, ui = cp ? uM : uO;
            d2v nr, ni;
            d2v w0, w1;     // the shadow vector: Re and Im of the two elements
            if constexpr (HASH) {
                uint64_t const hq = shadow_quad(key, uint32_t(lr), uint32_t(j), LN);   // rows 2 lr, 2 lr + 1 of column j
#pragma unroll
                for (int e = 0; e < 2; ++e) { w0[e] = shadow_pick(hq, e, 0); w1[e] = shadow_pick(hq, e, 1); }
            } else {
                f2v const wO = f2v{__shfl_xor(wM[0], 8), __shfl_xor(wM[1], 8)};
                w0 = cp ? d2v{wO[0], wO[1]} : d2v{wM[0], wM[1]}; w1 = cp ? d2v{wM[0], wM[1]} : d2v{wO[0], wO[1]};
            }
            if constexpr (EPI == EPI_XPAY_DOT) {          // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                d2v const vO = xor8(vM), vr = cp ? vO : vM, vi = cp ? vM : vO;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    R const tr = __builtin_fma(-si, ui[e], __builtin_fma(sr, ur[e], vr[e]));
                    R const ti = __builtin_fma(sr, ui[e], __builtin_fma(si, ur[e], vi[e]));
                    nr[e] = __builtin_fma(-si, ti, __builtin_fma(sr, tr, yr[e]));
                    ni[e] = __builtin_fma(sr, ti, __builtin_fma(si, tr, yi[e]));
                }
            } else {                                      // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    nr[e] = __builtin_fma(-si, yi[e], __builtin_fma(sr, yr[e], ur[e]));
                    ni[e] = __builtin_fma(sr, yi[e], __builtin_fma(si, yr[e], ui[e]));
                }
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {                 // every lane has both parts: the lanes of plane 0 are the ones that count
                double const dr = nr[e], di = ni[e];
                part[0] = __builtin_fma(-di, w1[e], __builtin_fma(dr, w0[e], part[0]));
                part[1] = __builtin_fma(di, w0[e], __builtin_fma(dr, w1[e], part[1]));
                if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[2]));
            }
            __builtin_nontemporal_store(yM, (d2v*)((R*)a.Y + yoff));
            __builtin_nontemporal_store(cp ? ni : nr, (d2v*)((R*)a.e0 + yoff));
        } else if constexpr (EPI == EPI_RESIDUAL) {       // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
            uint32_t const bq = a.bOfX ? a.bOfX[y] : y;
            d2v bM = d2v{0, 0};
            if (bq != 0xffffffffu) bM = *(d2v const*)((R const*)a.B + size_t(bq) * 2 * P + mine);
            d2v const bO = xor8(bM), br = cp ? bO : bM, bi = cp ? bM : bO;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                double const dr = yr[e] + R(-1) * br[e], di = yi[e] + R(-1) * bi[e];
                part[0] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[0]));
            }
        } else {
            __builtin_nontemporal_store(yM, (d2v*)((R*)a.Y + yoff));
        }
    }
    if constexpr (NPL > 0) {
        // the rows of a column sit 16 lanes apart (lr); lanes 0..7 (plane 0, lr 0) hold the column sums
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
            double v = part[p];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 8) s[wave][p][lane] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, jj = e % LN;
            double const sum = ((s[0][p][jj] + s[1][p][jj]) + s[2][p][jj]) + s[3][p][jj];
            write_record<EPI>(a, chunk, LN, p, jj, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 8 x 8 complex<double>, COLUMN-BATCHED (r03).  Block columns whose row patterns are identical (Plan::colBatch: dense right-hand-side columns, BASELINE
// config 5) are multiplied nb <= kColBatchMax = 2 at a time: the work group of chunk c of the FIRST column of a batch also does chunk c of the other columns -- same
// block rows, same A blocks, the X / Y blocks a column's block count further on -- so that an A block is fetched once for nb block products; the launch
// runs over the chunks of the batches' first columns only (DevPlan::orderB).  With blocks of 1 KiB the operand path bounds this shape (timing-only probe, profiles/r03_probes.txt:
// 3 of 4 A fetches skipped = -16 % / -22 % on the fused multiplies).  Chunks, records and every sum are those of k_spmm_ilv8: bit-identical results.
template <int EPI, bool HASH, int NB, bool FIRST = false>   // NB: columns of a batch at most; FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256, 3) void k_spmm_ilv8b(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    constexpr int LN = 8, P = 64, NPL = EpiPlanes<EPI>::N;
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15, cp = lc >> 3, j = lc & 7;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    uint32_t const cb = a.colBatch[col];
    if (cb & 15u) return;                       // (a later column of a batch: not in this launch's order, SpmmArgs::order = DevPlan::orderB)
    int const nb = int(cb >> 4);                // 1 ... NB columns
    uint32_t dBlk[NB], dChk[NB];                // how far the blocks / chunks of column col + k lie behind those of column col
    R sr[NB], si[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        dBlk[k] = 0; dChk[k] = 0; sr[k] = 0; si[k] = 0;
        if (k < nb) {
            dBlk[k] = a.colStart[col + k] - a.colStart[col]; dChk[k] = a.colChunkPtr[col + k] - a.colChunkPtr[col];
            if constexpr (UPD) { sr[k] = ((R const*)a.sc)[(size_t(col + k) * 2 + 0) * LN + j]; si[k] = ((R const*)a.sc)[(size_t(col + k) * 2 + 1) * LN + j]; }
        }
    }
    double part[NB][NPL > 0 ? NPL : 1] = {};
    __shared__ double s[NB][4][NPL > 0 ? NPL : 1][LN];

    int const mine = cp * P + (lr * 8 + j) * 2;                                 // this lane's 16 bytes of an X-shaped block
    R const* const A0 = (R const*)a.A + cp * P + (lr * 8 + 2 * (j & 3) + (j >> 2)) * 2;   // A: row pi(j) of plane cp, k pair lr
    R const* const X0 = (R const*)a.X + mine;
    struct Ops { d2v av; d2v xv[NB]; };
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        o.av = *(d2v const*)(A0 + size_t(pairs[2 * size_t(q)]) * 2 * P);
        uint32_t const xb = pairs[2 * size_t(q) + 1];
#pragma unroll
        for (int k = 0; k < NB; ++k) if (k < nb) o.xv[k] = *(d2v const*)(X0 + size_t(xb + dBlk[k]) * 2 * P);
    };
    for (uint32_t u = wave; u < last - first; u += 4) {
        uint32_t const y = first + u;
        T4 acc[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[k] = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], nq = starts[y + 1] - q0;
        constexpr int DEPTH = 2;   // block products in flight (3 | 4 measured level or slower, profiles/r03_column_batches.txt)
        Ops o[DEPTH];
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) if (uint32_t(dd) < nq) fetch(o[dd], q0 + dd);
        for (uint32_t base = 0; base < nq; base += DEPTH) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) {
                if (base + dd < nq) {
#pragma unroll
                    for (int k = 0; k < NB; ++k) if (k < nb) {
                        acc[k] = Acc<R>::mma(o[dd].av[0], o[dd].xv[k][0], acc[k]);
                        acc[k] = Acc<R>::mma(o[dd].av[1], o[dd].xv[k][1], acc[k]);
                    }
                    if (base + dd + DEPTH < nq) fetch(o[dd], q0 + base + dd + DEPTH);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) if (k < nb) {
            size_t const yoff = size_t(y + dBlk[k]) * 2 * P + mine;
            d2v uMk = d2v{0, 0}, vMk = d2v{0, 0}; f2v wMk = f2v{0, 0};
            if constexpr (UPD) {   // (requested here, not in front of the products: measured better with two columns per wave)
                if constexpr (!(EPI == EPI_XPAY_DOT && FIRST)) {   // (first iteration of a solve: old v4 = v8 = 0, not read)
                    uMk = __builtin_nontemporal_load((d2v const*)((R const*)a.e0 + yoff));
                    if constexpr (EPI == EPI_XPAY_DOT) vMk = __builtin_nontemporal_load((d2v const*)((R const*)a.e1 + yoff));
                }
                if constexpr (!HASH) wMk = __builtin_nontemporal_load((f2v const*)(a.v3 + yoff));
            }
            // lanes of plane 0 hold (Q00, Q10), lanes of plane 1 (Q01, Q11), rows 2 lr and 2 lr + 1: Re Y = Q00 - Q11, Im Y = Q01 + Q10
            d2v const qa = d2v{acc[k][0], acc[k][1]}, qb = xor8(d2v{acc[k][2], acc[k][3]});
            d2v const yM = cp ? d2v{qa[0] + qb[0], qa[1] + qb[1]} : d2v{qa[0] - qb[0], qa[1] - qb[1]};   // this lane's plane of Y
            d2v const yO = xor8(yM);
            d2v const yr = cp ? yO : yM, yi = cp ? yM : yO;
            if constexpr (UPD) {
                d2v const uO = xor8(uMk), ur = cp ? uO : uMk, ui = cp ? uMk : uO;
                d2v nr, ni;
                d2v w0, w1;     // the shadow vector: Re and Im of the two elements
                if constexpr (HASH) {
                    uint64_t const key = shadow_key(uint32_t(a.origCol[col + k]), a.rowI[y]);   // (the batch's columns have the same block rows)
                    uint64_t const hq = shadow_quad(key, uint32_t(lr), uint32_t(j), LN);       // rows 2 lr, 2 lr + 1 of column j
#pragma unroll
                    for (int e = 0; e < 2; ++e) { w0[e] = shadow_pick(hq, e, 0); w1[e] = shadow_pick(hq, e, 1); }
                } else {
                    f2v const wO = f2v{__shfl_xor(wMk[0], 8), __shfl_xor(wMk[1], 8)};
                    w0 = cp ? d2v{wO[0], wO[1]} : d2v{wMk[0], wMk[1]}; w1 = cp ? d2v{wMk[0], wMk[1]} : d2v{wO[0], wO[1]};
                }
                if constexpr (EPI == EPI_XPAY_DOT) {          // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                    d2v const vO = xor8(vMk), vr = cp ? vO : vMk, vi = cp ? vMk : vO;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        R const tr = __builtin_fma(-si[k], ui[e], __builtin_fma(sr[k], ur[e], vr[e]));
                        R const ti = __builtin_fma(sr[k], ui[e], __builtin_fma(si[k], ur[e], vi[e]));
                        nr[e] = __builtin_fma(-si[k], ti, __builtin_fma(sr[k], tr, yr[e]));
                        ni[e] = __builtin_fma(sr[k], ti, __builtin_fma(si[k], tr, yi[e]));
                    }
                } else {                                      // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        nr[e] = __builtin_fma(-si[k], yi[e], __builtin_fma(sr[k], yr[e], ur[e]));
                        ni[e] = __builtin_fma(sr[k], yi[e], __builtin_fma(si[k], yr[e], ui[e]));
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {                 // every lane has both parts: the lanes of plane 0 are the ones that count
                    double const dr = nr[e], di = ni[e];
                    part[k][0] = __builtin_fma(-di, w1[e], __builtin_fma(dr, w0[e], part[k][0]));
                    part[k][1] = __builtin_fma(di, w0[e], __builtin_fma(dr, w1[e], part[k][1]));
                    if constexpr (EPI == EPI_AXPY_NRM_DOT) part[k][2] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[k][2]));
                }
                __builtin_nontemporal_store(yM, (d2v*)((R*)a.Y + yoff));
                __builtin_nontemporal_store(cp ? ni : nr, (d2v*)((R*)a.e0 + yoff));
            } else if constexpr (EPI == EPI_RESIDUAL) {       // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
                uint32_t const bq = a.bOfX ? a.bOfX[y + dBlk[k]] : y + dBlk[k];
                d2v bM = d2v{0, 0};
                if (bq != 0xffffffffu) bM = *(d2v const*)((R const*)a.B + size_t(bq) * 2 * P + mine);
                d2v const bO = xor8(bM), br = cp ? bO : bM, bi = cp ? bM : bO;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    double const dr = yr[e] + R(-1) * br[e], di = yi[e] + R(-1) * bi[e];
                    part[k][0] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[k][0]));
                }
            } else {
                __builtin_nontemporal_store(yM, (d2v*)((R*)a.Y + yoff));
            }
        }
    }
    if constexpr (NPL > 0) {
        // the rows of a column sit 16 lanes apart (lr); lanes 0..7 (plane 0, lr 0) hold the column sums
#pragma unroll
        for (int k = 0; k < NB; ++k)
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                double v = part[k][p];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 8) s[k][wave][p][lane] = v;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < NB * NPL * LN; e += 256) {
            int const k = e / (NPL * LN), p = (e / LN) % NPL, jj = e % LN;
            if (k < nb) {
                double const sum = ((s[k][0][p][jj] + s[k][1][p][jj]) + s[k][2][p][jj]) + s[k][3][p][jj];
                uint32_t dc = dChk[0];
#pragma unroll
                for (int kk = 1; kk < NB; ++kk) if (kk == k) dc = dChk[kk];
                write_record<EPI>(a, chunk + dc, LN, p, jj, sum);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// 8 x 32 and 8 x 64 complex<double> on the row-pair-interleaved order: k_spmm_ilv8's tile ([Re A; Im A] x [Re X | Im X], one exchange
// with the lane 8 further) once per group of 8 block columns.  A wave-wide 16-byte access covers both planes and all 8 rows of ONE
// column group (8 segments of 128 bytes), so a block product is 1 + LN / 8 loads of 1 KiB (k_spmm_mfma8 on the native order: 2 + 2 LN / 8
// of 512 bytes) and every vector of the epilogue LN / 8 accesses per Y block.  No epilogue prefetch (the accumulators of 8 column groups
// are 64 registers), the shadow vector is read.
template <int LN, int EPI, bool FIRST = false>   // FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256) void k_spmm_ilv8w(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = double;
    static_assert(LN > 8 && (LN % 8 == 0 || LN < 16), "groups of 8 block columns, the last one of 8 x 9 | 8 x 10 ragged");
    constexpr int NTB = (LN + 7) / 8;                 // column groups of a block
    constexpr bool RAGGED = (LN % 8 != 0);            // (r04) 8 x 9, 8 x 10: the second group has 1 | 2 columns -- its other lanes load nothing, multiply zeros and store nothing
    constexpr int NT = (NTB > 4) ? 4 : NTB;           // column groups of one unit of work of a wave (8 x 64: a Y block is two units;
                                                      //  all 8 groups in one wave need 255-271 VGPRs = one wave per SIMD)
    constexpr int HALVES = NTB / NT;
    constexpr int P = 8 * LN, PA = 64, NPL = EpiPlanes<EPI>::N;
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    using T4 = d4;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15, cp = lc >> 3, j = lc & 7;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    double part[NPL > 0 ? NPL : 1][NT] = {};
    __shared__ double s[4][NPL > 0 ? NPL : 1][LN];
    if constexpr (NPL > 0) {   // (8 x 64: a wave meets both halves of the columns only if it has at least two units: clear what it may not write)
        for (int e = threadIdx.x; e < 4 * NPL * LN; e += 256) (&s[0][0][0])[e] = 0;
        __syncthreads();
    }

    R const* const A0 = (R const*)a.A + cp * PA + (lr * 8 + 2 * (j & 3) + (j >> 2)) * 2;   // A: row pi(j) of plane cp, k pair lr (as k_spmm_ilv8)
    struct Ops { d2v av, xv[NT]; };
    // units u = wave, wave + 4, ...: unit u is (Y block u / HALVES, half u % HALVES of its column groups); 4 is a multiple of HALVES,
    // so a wave keeps its half
    static_assert(4 % HALVES == 0, "a wave keeps its half of the column groups");
    int const t0 = (wave % HALVES) * NT;                  // first column group of this wave's units
    // this lane's 16 bytes of column group t0 + t of an X-shaped block: plane cp, row pair lr, column 8 (t0 + t) + j
    auto mine = [&](int t) { return cp * P + (lr * LN + 8 * (t0 + t) + j) * 2; };
    auto live = [&](int t) { return !RAGGED || 8 * (t0 + t) + j < LN; };      // this lane's column of group t exists
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        o.av = *(d2v const*)(A0 + size_t(pairs[2 * size_t(q)]) * 2 * PA);
        R const* Xb = (R const*)a.X + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
        for (int t = 0; t < NT; ++t) o.xv[t] = live(t) ? *(d2v const*)(Xb + mine(t)) : d2v{0, 0};
    };
    for (uint32_t u = wave; u < (last - first) * HALVES; u += 4) {
        uint32_t const y = first + u / HALVES;
        T4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = Acc<R>::mma(o.av[0], o.xv[t][0], acc[t]);
                acc[t] = Acc<R>::mma(o.av[1], o.xv[t][1], acc[t]);
            }
        };
        // (r04) with at most two column groups per wave (8 x 9, 8 x 10) the epilogue operands are requested in FRONT of the block products, as k_spmm_ilv8 does:
        // 20 registers; with four groups they are not (the accumulators of four groups are 32, the operand sets 40 registers)
        constexpr bool PRE = UPD && (NT <= 2);
        size_t const yb = size_t(y) * 2 * P;
        d2v uP[PRE ? NT : 1], vP[PRE ? NT : 1]; f2v wP[PRE ? NT : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uP[t] = d2v{0, 0}; vP[t] = d2v{0, 0}; wP[t] = f2v{0, 0};
                if (live(t)) {
                    if constexpr (!(EPI == EPI_XPAY_DOT && FIRST)) {
                        uP[t] = ld_stream<!RAGGED>((d2v const*)((R const*)a.e0 + yb + mine(t)));
                        if constexpr (EPI == EPI_XPAY_DOT) vP[t] = ld_stream<!RAGGED>((d2v const*)((R const*)a.e1 + yb + mine(t)));
                    }
                    wP[t] = ld_stream<!RAGGED>((f2v const*)(a.v3 + yb + mine(t)));
                }
            }
        }
        Ops o0, o1;
        if (q0 < q1) fetch(o0, q0);
        if (q0 + 1 < q1) fetch(o1, q0 + 1);
        uint32_t q = q0;
        for (; q + 2 <= q1; q += 2) {
            mma(o0);
            if (q + 2 < q1) fetch(o0, q + 2);
            mma(o1);
            if (q + 3 < q1) fetch(o1, q + 3);
        }
        if (q < q1) mma(o0);

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            size_t const yoff = yb + mine(t);
            bool const on = live(t);
            // lanes of plane 0 hold (Q00, Q10), lanes of plane 1 (Q01, Q11), rows 2 lr and 2 lr + 1: Re Y = Q00 - Q11, Im Y = Q01 + Q10
            d2v const qa = d2v{acc[t][0], acc[t][1]}, qb = xor8(d2v{acc[t][2], acc[t][3]});
            d2v const yM = cp ? d2v{qa[0] + qb[0], qa[1] + qb[1]} : d2v{qa[0] - qb[0], qa[1] - qb[1]};   // this lane's plane of Y
            d2v const yO = xor8(yM);
            d2v const yr = cp ? yO : yM, yi = cp ? yM : yO;
            if constexpr (UPD) {
                R const srt = on ? ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + 8 * (t0 + t) + j] : R(0);
                R const sit = on ? ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + 8 * (t0 + t) + j] : R(0);
                d2v uM = d2v{0, 0}, vM = d2v{0, 0}; f2v wM = f2v{0, 0};
                if constexpr (PRE) { uM = uP[t]; vM = vP[t]; wM = wP[t]; }
                else {
                    if constexpr (!(EPI == EPI_XPAY_DOT && FIRST)) {   // (first iteration of a solve: old v4 = v8 = 0, not read)
                        if (on) uM = ld_stream<!RAGGED>((d2v const*)((R const*)a.e0 + yoff));
                        if constexpr (EPI == EPI_XPAY_DOT) if (on) vM = ld_stream<!RAGGED>((d2v const*)((R const*)a.e1 + yoff));
                    }
                    if (on) wM = ld_stream<!RAGGED>((f2v const*)(a.v3 + yoff));
                }
                d2v const uO = xor8(uM), ur = cp ? uO : uM, ui = cp ? uM : uO;
                f2v const wO = f2v{__shfl_xor(wM[0], 8), __shfl_xor(wM[1], 8)};
                d2v const w0 = cp ? d2v{wO[0], wO[1]} : d2v{wM[0], wM[1]}, w1 = cp ? d2v{wM[0], wM[1]} : d2v{wO[0], wO[1]};
                d2v nr, ni;
                if constexpr (EPI == EPI_XPAY_DOT) {          // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                    d2v const vO = xor8(vM), vr = cp ? vO : vM, vi = cp ? vM : vO;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        R const tr = __builtin_fma(-sit, ui[e], __builtin_fma(srt, ur[e], vr[e]));
                        R const ti = __builtin_fma(srt, ui[e], __builtin_fma(sit, ur[e], vi[e]));
                        nr[e] = __builtin_fma(-sit, ti, __builtin_fma(srt, tr, yr[e]));
                        ni[e] = __builtin_fma(srt, ti, __builtin_fma(sit, tr, yi[e]));
                    }
                } else {                                      // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        nr[e] = __builtin_fma(-sit, yi[e], __builtin_fma(srt, yr[e], ur[e]));
                        ni[e] = __builtin_fma(srt, yi[e], __builtin_fma(sit, yr[e], ui[e]));
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {                 // every lane has both parts: the lanes of plane 0 are the ones that count
                    double const dr = nr[e], di = ni[e];
                    part[0][t] = __builtin_fma(-di, w1[e], __builtin_fma(dr, w0[e], part[0][t]));
                    part[1][t] = __builtin_fma(di, w0[e], __builtin_fma(dr, w1[e], part[1][t]));
                    if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2][t] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[2][t]));
                }
                if (on) {
                    st_stream<!RAGGED>((d2v*)((R*)a.Y + yoff), yM);
                    st_stream<!RAGGED>((d2v*)((R*)a.e0 + yoff), cp ? ni : nr);
                }
            } else if constexpr (EPI == EPI_RESIDUAL) {       // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
                d2v bM = d2v{0, 0};
                if (bq != 0xffffffffu && on) bM = *(d2v const*)((R const*)a.B + size_t(bq) * 2 * P + mine(t));
                d2v const bO = xor8(bM), br = cp ? bO : bM, bi = cp ? bM : bO;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    double const dr = yr[e] + R(-1) * br[e], di = yi[e] + R(-1) * bi[e];
                    part[0][t] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[0][t]));
                }
            } else {
                if (on) st_stream<!RAGGED>((d2v*)((R*)a.Y + yoff), yM);
            }
        }
    }
    if constexpr (NPL > 0) {
        // the rows of a column sit 16 lanes apart (lr); lanes 0..7 (plane 0, lr 0) hold the column sums of their column group
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double v = part[p][t];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 8 && (!RAGGED || 8 * (t0 + t) + lane < LN)) s[wave][p][8 * (t0 + t) + lane] = v;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, jj = e % LN;
            double const sum = ((s[0][p][jj] + s[1][p][jj]) + s[2][p][jj]) + s[3][p][jj];
            write_record<EPI>(a, chunk, LN, p, jj, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 8 x 8, 8 x 32 and 8 x 64 complex<float> with groups of FOUR rows interleaved (plane[r/4][s][r%4], the order of k_spmm_ilv16f): the float
// counterpart of k_spmm_ilv8 / k_spmm_ilv8w.  The tile is again [Re A; Im A] (16 x 8) x [Re X | Im X] (8 x 16) per group of 8 block columns, but
// a 16-byte access of a lane is a k QUAD, and a block of 8 rows has only two of them where v_mfma_f32_16x16x4_f32 has four k slots per
// step: the upper two slots take the NEXT block product of the same Y block (lane groups 0, 1: product q, lane groups 2, 3: product q + 1;
// both sums land in the same accumulators), so ONE wave-wide 1-KiB access fetches the A blocks of two products, one per column group their X
// blocks (k_spmm_mfma8<float> on the native order: 4 bytes per lane and access).  Accumulator registers 0..3 of a lane are the rows
// 4 (lane / 16) .. + 3 of [Re A; Im A] X: lane groups 0, 1 hold the Re A part, groups 2, 3 the Im A part; Re Y = Q00 - Q11 and Im Y = Q01 + Q10
// meet through one exchange with lane ^ 40 (other lane-group half, other plane), after which lanes 0 .. 31 hold one 16-byte piece of Y each
// (quad lane / 16 of column lane % 8, plane (lane % 16) / 8) -- the Y block, every epilogue operand and every result of a column group
// is one half-wave access.  The shadow vector is read.
template <int LN, int EPI, bool FIRST = false>   // FIRST: the launch of the first iteration of a solve (SpmmArgs::first)
__global__ __launch_bounds__(256) void k_spmm_ilv8f(SpmmArgs a) {
    if (gate_closed(a)) return;
    using R = float;
    static_assert(LN % 8 == 0, "groups of 8 block columns");
    constexpr int NTB = LN / 8;                       // column groups of a block
    constexpr int NT = (NTB > 4) ? 4 : NTB;           // column groups of one unit of work of a wave
    constexpr int HALVES = NTB / NT;
    constexpr int P = 8 * LN, PA = 64, NPL = EpiPlanes<EPI>::N;
    constexpr bool UPD = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15, cp = lc >> 3, j = lc & 7;
    int const g = lr & 1;                             // k quad (operands) | row quad (results, lanes 0 .. 31)
    bool const second = (lr >= 2);                    // operands: this lane feeds the second product of a pair
    bool const owner = (lr < 2);                      // results: lanes 0 .. 31 own the pieces of Y
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;
    uint32_t const first = a.chunkFirst[chunk], last = a.chunkFirst[chunk + 1], col = a.chunkCol[chunk];
    double part[NPL > 0 ? NPL : 1][NT] = {};
    __shared__ double s[4][NPL > 0 ? NPL : 1][LN];
    if constexpr (NPL > 0 && HALVES > 1) {   // a wave writes the sums of its half of the column groups only
        for (int e = threadIdx.x; e < 4 * NPL * LN; e += 256) (&s[0][0][0])[e] = 0;
        __syncthreads();
    }
    static_assert(4 % HALVES == 0, "a wave keeps its half of the column groups");
    int const t0 = (wave % HALVES) * NT;              // first column group of this wave's units
    // 16 bytes of an X-shaped block: plane cp, quad g, column 8 (t0 + t) + j;  of an A block (transposed): plane cp, k quad g, row j
    auto mine = [&](int t) { return cp * P + (g * LN + 8 * (t0 + t) + j) * 4; };
    int const mineA = cp * PA + (g * 8 + j) * 4;
    struct Ops { f4v av, xv[NT]; };
    auto fetch = [&](Ops& o, uint32_t q, uint32_t q1) __attribute__((always_inline)) {   // products q (lane groups 0, 1) and q + 1 (2, 3)
        bool const two = (q + 1 < q1);
        uint32_t const ia0 = pairs[2 * size_t(q)], ix0 = pairs[2 * size_t(q) + 1];
        uint32_t const ia1 = two ? pairs[2 * size_t(q) + 2] : ia0, ix1 = two ? pairs[2 * size_t(q) + 3] : ix0;
        uint32_t const ia = second ? ia1 : ia0, ix = second ? ix1 : ix0;
        o.av = f4v{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NT; ++t) o.xv[t] = f4v{0, 0, 0, 0};
        if (!second || two) {
            o.av = *(f4v const*)((R const*)a.A + size_t(ia) * 2 * PA + mineA);
            R const* Xb = (R const*)a.X + size_t(ix) * 2 * P;
#pragma unroll
            for (int t = 0; t < NT; ++t) o.xv[t] = *(f4v const*)(Xb + mine(t));
        }
    };
    for (uint32_t u = wave; u < (last - first) * HALVES; u += 4) {
        uint32_t const y = first + u / HALVES;
        f4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        auto mma = [&](Ops const& o) __attribute__((always_inline)) {
#pragma unroll
            for (int e = 0; e < 4; ++e)                   // step e contracts k = 4 g + e of product q (slots 0, 1) and q + 1 (slots 2, 3)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = Acc<R>::mma(o.av[e], o.xv[t][e], acc[t]);
        };
        Ops o0, o1;
        if (q0 < q1) fetch(o0, q0, q1);
        if (q0 + 2 < q1) fetch(o1, q0 + 2, q1);
        uint32_t q = q0;
        for (; q + 4 <= q1 + 1 && q + 2 < q1; q += 4) {   // two pairs per trip while a second pair exists
            mma(o0);
            if (q + 4 < q1) fetch(o0, q + 4, q1);
            mma(o1);
            if (q + 6 < q1) fetch(o1, q + 6, q1);
        }
        if (q < q1) mma(o0);

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
        size_t const yb = size_t(y) * 2 * P;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // lane (lr, cp): lr < 2: [Re A X] rows of quad lr, x (Re | Im) X; lr >= 2: [Im A X] rows of quad lr - 2.  Partner lane ^ 40.
            f4v const v = f4v{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
            f4v const o = f4v{__shfl_xor(v[0], 40), __shfl_xor(v[1], 40), __shfl_xor(v[2], 40), __shfl_xor(v[3], 40)};
            // owners: plane 0: Re Y = Q00 - Q11, plane 1: Im Y = Q01 + Q10
            f4v const yM = cp ? f4v{v[0] + o[0], v[1] + o[1], v[2] + o[2], v[3] + o[3]} : f4v{v[0] - o[0], v[1] - o[1], v[2] - o[2], v[3] - o[3]};
            f4v const yO = f4v{__shfl_xor(yM[0], 8), __shfl_xor(yM[1], 8), __shfl_xor(yM[2], 8), __shfl_xor(yM[3], 8)};   // the other plane of the same elements
            f4v const yr = cp ? yO : yM, yi = cp ? yM : yO;
            size_t const yoff = yb + mine(t);
            if constexpr (UPD) {
                R const srt = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + 8 * (t0 + t) + j];
                R const sit = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + 8 * (t0 + t) + j];
                f4v uM = f4v{0, 0, 0, 0}, vM = f4v{0, 0, 0, 0}, wM = f4v{0, 0, 0, 0};
                if (owner) {
                    if constexpr (!(EPI == EPI_XPAY_DOT && FIRST)) {   // (first iteration of a solve: old v4 = v8 = 0, not read)
                        uM = __builtin_nontemporal_load((f4v const*)((R const*)a.e0 + yoff));
                        if constexpr (EPI == EPI_XPAY_DOT) vM = __builtin_nontemporal_load((f4v const*)((R const*)a.e1 + yoff));
                    }
                    wM = __builtin_nontemporal_load((f4v const*)(a.v3 + yoff));
                }
                auto x8 = [](f4v z) { return f4v{__shfl_xor(z[0], 8), __shfl_xor(z[1], 8), __shfl_xor(z[2], 8), __shfl_xor(z[3], 8)}; };
                f4v const uO = x8(uM), ur = cp ? uO : uM, ui = cp ? uM : uO;
                f4v const wO = x8(wM), w0 = cp ? wO : wM, w1 = cp ? wM : wO;
                f4v nr, ni;
                if constexpr (EPI == EPI_XPAY_DOT) {          // v9 := A v6; v4 := v8 + beta v4; v4 := v9 + beta v4 (tfqmrgpu_core.hxx:196-202)
                    f4v const vO = x8(vM), vr = cp ? vO : vM, vi = cp ? vM : vO;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        R const tr = __builtin_fmaf(-sit, ui[e], __builtin_fmaf(srt, ur[e], vr[e]));
                        R const ti = __builtin_fmaf(srt, ui[e], __builtin_fmaf(sit, ur[e], vi[e]));
                        nr[e] = __builtin_fmaf(-sit, ti, __builtin_fmaf(srt, tr, yr[e]));
                        ni[e] = __builtin_fmaf(srt, ti, __builtin_fmaf(sit, tr, yi[e]));
                    }
                } else {                                      // v8 := A v6; v5 := alfa v8 + v5 (tfqmrgpu_core.hxx:224-228)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        nr[e] = __builtin_fmaf(-sit, yi[e], __builtin_fmaf(srt, yr[e], ur[e]));
                        ni[e] = __builtin_fmaf(srt, yi[e], __builtin_fmaf(sit, yr[e], ui[e]));
                    }
                }
                if (owner) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {             // every owner lane has both parts: the lanes of plane 0 are the ones that count
                        double const dr = nr[e], di = ni[e], x0 = w0[e], x1 = w1[e];
                        part[0][t] = __builtin_fma(-di, x1, __builtin_fma(dr, x0, part[0][t]));
                        part[1][t] = __builtin_fma(di, x0, __builtin_fma(dr, x1, part[1][t]));
                        if constexpr (EPI == EPI_AXPY_NRM_DOT) part[2][t] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[2][t]));
                    }
                    __builtin_nontemporal_store(yM, (f4v*)((R*)a.Y + yoff));
                    __builtin_nontemporal_store(cp ? ni : nr, (f4v*)((R*)a.e0 + yoff));
                }
            } else if constexpr (EPI == EPI_RESIDUAL) {       // |A x - b|^2, nothing stored (tfqmrgpu_core.hxx:265-269)
                f4v bM = f4v{0, 0, 0, 0};
                if (owner && bq != 0xffffffffu) bM = *(f4v const*)((R const*)a.B + size_t(bq) * 2 * P + mine(t));
                f4v const bO = f4v{__shfl_xor(bM[0], 8), __shfl_xor(bM[1], 8), __shfl_xor(bM[2], 8), __shfl_xor(bM[3], 8)};
                f4v const br = cp ? bO : bM, bi = cp ? bM : bO;
                if (owner) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        R const rr = yr[e] + R(-1) * br[e], ri = yi[e] + R(-1) * bi[e];
                        double const dr = rr, di = ri;
                        part[0][t] = __builtin_fma(di, di, __builtin_fma(dr, dr, part[0][t]));
                    }
                }
            } else {
                if (owner) __builtin_nontemporal_store(yM, (f4v*)((R*)a.Y + yoff));
            }
        }
    }
    if constexpr (NPL > 0) {
        // owner lanes of plane 0: lane j of lane group 0 | 1 holds the sums of quad 0 | 1 of column 8 (t0 + t) + j
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double v = part[p][t];
                v += __shfl_xor(v, 16);
                if (lane < 8) s[wave][p][8 * (t0 + t) + lane] = v;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, jj = e % LN;
            double const sum = ((s[0][p][jj] + s[1][p][jj]) + s[2][p][jj]) + s[3][p][jj];
            write_record<EPI>(a, chunk, LN, p, jj, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// MFMA kernel for 8-row blocks (LM == 8, LN % 8 == 0).  A 16x16 tile would be half empty, so the tile is
// filled with the complex structure instead:   [Re A]             [Re A Re X | Re A Im X]
//                                               [Im A] (16 x 8)  x  [Re X | Im X] (8 x 16)  =  [Im A Re X | Im A Im X]
// i.e. all four real products of one 8x8 complex block product come out of ONE accumulator tile with
// K = 8 -> 2 MFMAs (every flop useful).  The native layouts again are the operand layouts: lane l feeds
// A[c = (l%16)/8][k0 + l/16][(l%16)%8] and X[c = (l%16)/8][k0 + l/16][8 nt + (l%16)%8].
// After the pair loop the tile goes through a wave-private LDS patch and comes back as one complex
// element per lane:  Y = (Q00 - Q11) + i (Q01 + Q10),  lane l <-> element (row l/8, column l%8).
template <typename R, int LM, int LN, int EPI, bool PRE>
__global__ __launch_bounds__(256) void k_spmm_mfma8(SpmmArgs a) {
    if (gate_closed(a)) return;
    static_assert(LM == 4 || LM == 8, "[Re A; Im A] must fit the 16 rows of a tile");
    constexpr int P = LM * LN, NT = (LN + 7) / 8;   // LN = 5, 9, 10: the last tile has 5, 1 or 2 columns, the rest is masked
    constexpr int KS = LM / 4;                       // MFMA k-steps per block product
    constexpr int NPL = EpiPlanes<EPI>::N;
    constexpr bool RAGGED = (LN % 8 != 0);
    using T4 = typename Acc<R>::T;
    __shared__ R tile[4][16][17];                      // one patch per wave, padded rows
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    int const part8 = lc >> 3, j8 = lc & 7;            // X operand: plane and column inside the tile
    int const pa = lc / LM, ia = lc % LM;              // A operand: plane (LM == 4: lanes with pa >= 2 feed zeros) and row
    int const ei = lane >> 3, ej = lane & 7;           // epilogue side: element (ei, ej) of the LM x 8 tile (ei < LM)
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;   // XCD-aware launch order (tfq_plan.cpp)
    uint32_t first, last, col = 0;
    if (a.chunkFirst) { first = a.chunkFirst[chunk]; last = a.chunkFirst[chunk + 1]; col = a.chunkCol[chunk]; }
    else { first = chunk * a.CH; last = min(first + a.CH, a.nY); }

    R sr[NT], si[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { sr[nt] = 0; si[nt] = 0; }
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) if (!RAGGED || nt * 8 + ej < LN) {
            sr[nt] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + nt * 8 + ej];
            si[nt] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + nt * 8 + ej];
        }
    }
    double part[NPL > 0 ? NPL : 1][NT] = {};

    // (row ranges and index pairs are the same for every lane of a wave: scalar loads from the constant address space -- as plain global
    //  loads the index pair of a product was waited for with s_waitcnt vmcnt(0) right in front of its operand requests, which drained the
    //  DEPTH products "in flight" every time; r04, profiles/r04_four_row_shapes.txt)
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts;
    struct Ops { R a[KS]; R x[KS][NT]; };
    R const* const A0 = (R const*)a.A + (pa & 1) * (LM * LM) + ia;   // + k*LM
    R const* const X0 = (R const*)a.X + part8 * P + j8;              // + k*LN + nt*8
    for (uint32_t y = first + wave; y < last; y += 4) {
        T4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = T4{0, 0, 0, 0};
        uint32_t const q0 = starts[y], q1 = starts[y + 1];
        // the operands of this block's epilogue travel while its products are computed (PRE; TFQMRGPU_EPI_PREFETCH=0: behind them)
        EpiElem<R, EPI, LN == 8> eo[PRE ? NT : 1];
        if constexpr (PRE) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if ((LM == 8 || ei < LM) && (!RAGGED || nt * 8 + ej < LN))
                    eo[nt].load(a, size_t(y) * 2 * P + ei * LN + nt * 8 + ej, P);
        }
        auto fetch = [&](Ops& o, uint32_t q) {
            R const* Ab = A0 + size_t(pairs[2 * size_t(q)]) * 2 * LM * LM;
            R const* Xb = X0 + size_t(pairs[2 * size_t(q) + 1]) * 2 * P;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                int const k = 4 * s + lr;
                o.a[s] = (LM == 8 || pa < 2) ? Ab[k * LM] : R(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) o.x[s][nt] = (!RAGGED || nt * 8 + j8 < LN) ? Xb[k * LN + nt * 8] : R(0);
            }
        };
        auto mma = [&](Ops const& o) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = Acc<R>::mma(o.a[s], o.x[s][nt], acc[nt]);
        };
        // the block products are tiny (2 MFMAs, 2 KiB of operands): keep DEPTH of them in flight
        constexpr int DEPTH = (NT <= 2) ? 4 : 2;
        Ops o[DEPTH];
        uint32_t const nq = q1 - q0;
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) if (uint32_t(dd) < nq) fetch(o[dd], q0 + dd);
        for (uint32_t base = 0; base < nq; base += DEPTH) {
#pragma unroll
            for (int dd = 0; dd < DEPTH; ++dd) {
                if (base + dd < nq) {
                    mma(o[dd]);
                    if (base + dd + DEPTH < nq) fetch(o[dd], q0 + base + dd + DEPTH);
                }
            }
        }

        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[wave][Acc<R>::row(lane, r)][lc] = acc[nt][r];
            __builtin_amdgcn_wave_barrier();           // LDS operations of one wave complete in order
            int const er = ei % LM;                    // LM == 4: the upper half of the lanes has no element
            R const yr = tile[wave][er][ej] - tile[wave][er + LM][ej + 8];
            R const yi = tile[wave][er][ej + 8] + tile[wave][er + LM][ej];
            int const e = ei * LN + nt * 8 + ej;
            double accp[NPL > 0 ? NPL : 1] = {};
            if ((LM == 8 || ei < LM) && (!RAGGED || nt * 8 + ej < LN)) {   // STREAM for LN == 8: the tile is one contiguous plane
                if constexpr (PRE) epilogue_apply<R, EPI, LN == 8>(a, size_t(y) * 2 * P + e, P, yr, yi, sr[nt], si[nt], eo[nt], bq, e, accp);
                else epilogue<R, EPI, LN == 8>(a, size_t(y) * 2 * P + e, P, yr, yi, sr[nt], si[nt], bq, e, accp);
            }
#pragma unroll
            for (int p = 0; p < NPL; ++p) part[p][nt] += accp[p];
        }
    }

    if constexpr (NPL > 0) {
        // the rows of a column sit 8 lanes apart: add them, then the four waves in order
        __shared__ double s[4][NPL][LN];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double v = part[p][nt];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if (lane < 8 && (!RAGGED || nt * 8 + lane < LN)) s[wave][p][nt * 8 + lane] = v;
            }
        __syncthreads();
        for (int e = threadIdx.x; e < NPL * LN; e += 256) {
            int const p = e / LN, j = e % LN;
            double const sum = ((s[0][p][j] + s[1][p][j]) + s[2][p][j]) + s[3][p][j];
            write_record<EPI>(a, chunk, LN, p, j, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 4-row blocks that are too small for the tile kernel (4 x 4, and the float 4-row shapes: a block is 128 ... 1024 bytes).
// A thread group of 16, 20, 32 or 64 lanes owns one sub-block of 4 x min(LN, 16) elements, one element per lane; the operands
// of a block product are read ONCE per group (4 memory instructions per wave and product instead of 16 per lane in
// k_spmm_direct), pass through a group-private LDS patch and are broadcast from there.  Groups never straddle a wave and
// LDS operations of one wave complete in order, so no barrier is needed inside the product loop.
template <typename R, int LN, int EPI>
__global__ __launch_bounds__(256) void k_spmm_small4(SpmmArgs a) {
    if (gate_closed(a)) return;
    constexpr int LM = 4, P = LM * LN;
    constexpr int LNS = (LN > 16) ? 16 : LN;             // columns of a sub-block
    constexpr int NSUB = LN / LNS;                       // sub-blocks per block (LN = 32: 2)
    constexpr int PE = LM * LNS;                         // elements of a sub-block: 16, 20, 32, 64
    constexpr int GPW = 64 / PE;                         // thread groups per wave, PE lanes each: 4, 3 (4 x 5: lanes 60..63 idle), 2, 1
    constexpr int NG = 4 * GPW;                          // thread groups per work group
    constexpr int NPL = EpiPlanes<EPI>::N;
    static_assert(LN % LNS == 0 && NG % NSUB == 0, "a thread group keeps its sub-block index");
    // (r04: the patches of a wave's groups are read by ONE LDS instruction; at their natural strides -- 128 | 256 bytes in float -- the groups' segments share
    //  banks: padded by 16 bytes, as in k_spmm_s4w: 4 x 4 c iteration -2.7 %, 4 x 5 c -1 %; in double (4 x 5 z) a pad measured 0.4 % slower: none)
    constexpr int PADR = (sizeof(R) == 4) ? 4 : 0;
    __shared__ R As[NG][2 * LM * LM + PADR];
    __shared__ R Xs[NG][2 * PE + PADR];
    int const t = threadIdx.x, wv = t >> 6, ln = t & 63;
    bool const valid = (ln < GPW * PE);
    int const g = wv * GPW + (valid ? ln / PE : GPW - 1), e = valid ? ln % PE : PE;   // idle lanes walk with the wave's last group and touch nothing
    int const i = valid ? e / LNS : 0, jj = valid ? e % LNS : 0;
    int const j = (g % NSUB) * LNS + jj;                 // block column of this lane
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;   // XCD-aware launch order (tfq_plan.cpp)
    uint32_t first, last, col = 0;
    if (a.chunkFirst) { first = a.chunkFirst[chunk]; last = a.chunkFirst[chunk + 1]; col = a.chunkCol[chunk]; }
    else { first = chunk * a.CH; last = min(first + a.CH, a.nY); }

    R sr = 0, si = 0;
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
        sr = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + j];
        si = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + j];
    }
    double part[NPL > 0 ? NPL : 1] = {};

    // Three dependent requests lead to a product (row range -> index pairs -> operands).  As in k_spmm_m4 the work group fetches
    // the row ranges and index pairs of its whole chunk into LDS first, and a thread group requests the operands of up to NB
    // products before it multiplies the first: one memory latency per NB products instead of two per product.
    constexpr uint32_t kRows = 256, kPairs = 2048;       // (a float chunk of 4 x 4 blocks has 128 rows)
    constexpr int NB = (sizeof(R) == 8) ? 6 : 8;
    __shared__ uint32_t sStarts[kRows + 1];
    __shared__ uint32_t sPairs[2 * kPairs];
    uint32_t const nRows = last - first, nItems = nRows * NSUB;   // item = sub-block of a Y block; item % NSUB == g % NSUB
    uint32_t const qBase = a.starts[first], qEnd = a.starts[last];
    bool const inLds = (nRows <= kRows) && (qEnd - qBase <= kPairs);
    if (inLds) {
        for (uint32_t i = t; i <= nRows; i += 256) sStarts[i] = a.starts[first + i];
        for (uint32_t i = t; i < 2 * (qEnd - qBase); i += 256) sPairs[i] = a.pairs[2 * size_t(qBase) + i];
    }
    __syncthreads();

    auto multiply = [&](R const (&pa)[2], R const (&px)[2], R& yr, R& yi) __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();
        if (e < LM * LM) { As[g][e] = pa[0]; As[g][LM * LM + e] = pa[1]; }
        if (valid) { Xs[g][e] = px[0]; Xs[g][PE + e] = px[1]; }
        __builtin_amdgcn_wave_barrier();
        R cr = 0, ci = 0;
#pragma unroll
        for (int k = 0; k < LM; ++k) {
            R const ar = As[g][k * LM + i], ai = As[g][LM * LM + k * LM + i];
            R const xr = Xs[g][k * LNS + jj], xi = Xs[g][PE + k * LNS + jj];
            cr = fma_(-ai, xi, fma_(ar, xr, cr));
            ci = fma_(ai, xr, fma_(ar, xi, ci));
        }
        yr += cr; yi += ci;
    };

    for (uint32_t it = g; it < nItems; it += NG) {
        uint32_t const kr = it / NSUB, y = first + kr;
        int const eb = i * LN + j;
        size_t const off = size_t(y) * 2 * P + eb;
        EpiElem<R, EPI, false> eo;
        if (valid) eo.load(a, off, P);
        R yr = 0, yi = 0;
        if (inLds) {
            uint32_t const q0 = sStarts[kr] - qBase, q1 = sStarts[kr + 1] - qBase;
            for (uint32_t qb = q0; qb < q1; qb += NB) {
                R pa[NB][2], px[NB][2];
                uint32_t ia[NB], ix[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {           // (unconditional, inside the patch: all NB reads in flight at once)
                    uint32_t const qc = min(qb + u, kPairs - 1);
                    ia[u] = sPairs[2 * qc]; ix[u] = sPairs[2 * qc + 1];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    pa[u][0] = 0; pa[u][1] = 0; px[u][0] = 0; px[u][1] = 0;
                    if (qb + u < q1) {
                        R const* Ab = (R const*)a.A + size_t(ia[u]) * 2 * (LM * LM);
                        R const* Xb = (R const*)a.X + size_t(ix[u]) * 2 * P;
                        if (e < LM * LM) { pa[u][0] = Ab[e]; pa[u][1] = Ab[LM * LM + e]; }
                        if (valid) { px[u][0] = Xb[i * LN + j]; px[u][1] = Xb[P + i * LN + j]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < NB; ++u)
                    if (qb + u < q1) multiply(pa[u], px[u], yr, yi);
            }
        } else {   // a chunk whose index data exceed the LDS patch: one product in flight, indices from global memory
            uint32_t const q0 = a.starts[y], q1 = a.starts[y + 1];
            R pa[2] = {0, 0}, px[2] = {0, 0};                // operands of the next product, in flight
            auto fetch = [&](uint32_t q) __attribute__((always_inline)) {
                R const* Ab = (R const*)a.A + size_t(a.pairs[2 * size_t(q)]) * 2 * (LM * LM);
                R const* Xb = (R const*)a.X + size_t(a.pairs[2 * size_t(q) + 1]) * 2 * P;
                if (e < LM * LM) { pa[0] = Ab[e]; pa[1] = Ab[LM * LM + e]; }
                if (valid) { px[0] = Xb[i * LN + j]; px[1] = Xb[P + i * LN + j]; }
            };
            if (q0 < q1) fetch(q0);
            for (uint32_t q = q0; q < q1; ++q) {
                R const ca[2] = {pa[0], pa[1]}, cx[2] = {px[0], px[1]};
                if (q + 1 < q1) fetch(q + 1);
                multiply(ca, cx, yr, yi);
            }
        }
        if (valid) {
            uint32_t bq = 0xffffffffu;
            if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
            epilogue_apply<R, EPI, false>(a, off, P, yr, yi, sr, si, eo, bq, eb, part);
        }
    }

    if constexpr (NPL > 0) {
        // threads that share a block column: groups with the same sub-block index, 4 rows each; added in a fixed order
        __shared__ double red[NPL][256];
#pragma unroll
        for (int p = 0; p < NPL; ++p) red[p][t] = valid ? part[p] : 0.0;
        __syncthreads();
        for (int x = t; x < NPL * LN; x += 256) {
            int const p = x / LN, jx = x % LN;
            double sum = 0;
            // (not unrolled: with 4 columns and three records the compiler unrolled all 64 terms of a sum and held them in registers --
            //  134 VGPRs for k_spmm_small4<., 4, EPI_AXPY_NRM_DOT> against 76 for its siblings, half the waves per SIMD; r03)
#pragma unroll 1
            for (int gg = jx / LNS; gg < NG; gg += NSUB)
                for (int r = 0; r < LM; ++r) sum += red[p][(gg / GPW) * 64 + (gg % GPW) * PE + r * LNS + jx % LNS];
            write_record<EPI>(a, chunk, LN, p, jx, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 4-row blocks in double whose columns come in fours: four 4 x 4 x 4 products per v_mfma_f64_4x4x4_4b_f64.  The instruction
// keeps its four blocks interleaved at 4 lanes (measured with one-hot operands, scripts/mfma4_probe.hip): with lo = lane % 4,
// b = lane / 4 % 4, hi = lane / 16 a lane holds  A_b[i = lo][k = hi],  B_b[k = hi][j = lo]  and receives  D_b[i = hi][j = lo].
// A blocks are stored as [k][i] and X, Y blocks as [i][j], so a lane loads and stores its elements straight from the planes,
// at hi * 4 + lo (A) and hi * LN + its column(s) (X, Y): no LDS patch, no broadcast -- k_spmm_small4 spends 25 LDS
// instructions per Y block on them.  Where LN is a multiple of 8 a lane keeps W = 2 NEIGHBOURING columns (the 4 x 4
// products of the even and of the odd columns of an octet: which four columns share a product is free), so that X and
// every epilogue vector move as 16-byte accesses -- the memory pipe retires one wave-wide access per 16 clocks whatever
// its width (scripts/ta_rate.hip), and these kernels are bound by that rate (profiles/r04_small_shapes.txt).
// A slot (b of a wave, 16 per work group) walks over Y sub-blocks of 4 x 4 W columns; the four slots of a wave step
// together, a slot that has run out of products feeds zeros.
#ifndef TFQ_M4_NB
#define TFQ_M4_NB 8
#endif
template <int LN, int EPI>
__global__ __launch_bounds__(256) void k_spmm_m4(SpmmArgs a) {
    using R = double;
    if (gate_closed(a)) return;
    constexpr int W = (LN % 8 == 0) ? 2 : 1;             // neighbouring columns of a lane
    constexpr int LM = 4, P = LM * LN, CQ = 4 * W, NSUB = LN / CQ, NS = 16, NB = TFQ_M4_NB / W;
    constexpr int NPL = EpiPlanes<EPI>::N;
    static_assert(LN % CQ == 0 && NS % NSUB == 0, "a slot keeps its column group");
    int const t = threadIdx.x, wv = t >> 6, lane = t & 63;
    int const lo = lane & 3, b = (lane >> 2) & 3, hi = lane >> 4;
    int const slot = wv * 4 + b;
    int const j0 = (slot % NSUB) * CQ + W * lo;          // first block column of this lane (X and Y)
    int const ea = hi * LM + lo, ex = hi * LN + j0;      // this lane's element of an A block, its first of an X or Y block
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;   // XCD-aware launch order (tfq_plan.cpp)
    uint32_t first, last, col = 0;
    if (a.chunkFirst) { first = a.chunkFirst[chunk]; last = a.chunkFirst[chunk + 1]; col = a.chunkCol[chunk]; }
    else { first = chunk * a.CH; last = min(first + a.CH, a.nY); }

    R sr[W], si[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { sr[w] = 0; si[w] = 0; }
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            sr[w] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + j0 + w];
            si[w] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + j0 + w];
        }
    }
    double part[NPL > 0 ? NPL : 1][W] = {};

    // An item = a 4 x CQ sub-block of a Y block (item % NSUB == slot % NSUB); slot s takes items s, s + 16, ...  Three dependent
    // requests lead to a product (row range -> index pairs -> operands).  The work group fetches the row ranges and the index
    // pairs of its whole chunk into LDS first (two latencies, once), so that a trip -- up to NB products of four items per
    // wave -- waits for ONE memory latency; k_spmm_small4 waits for two per product.
    constexpr uint32_t kRows = 256, kPairs = 1024;       // LDS patch: chunks of at most 256 Y blocks (tfq_plan.cpp: 16 KiB of 256-byte blocks = 64)
    __shared__ uint32_t sStarts[kRows + 1];
    __shared__ uint32_t sPairs[2 * kPairs];
    uint32_t const nRows = last - first, nItems = nRows * NSUB;
    uint32_t const qBase = a.starts[first], qEnd = a.starts[last];      // (uniform: scalar loads)
    bool const inLds = (nRows <= kRows) && (qEnd - qBase <= kPairs);
    if (inLds) {
        for (uint32_t i = t; i <= nRows; i += 256) sStarts[i] = a.starts[first + i];
        for (uint32_t i = t; i < 2 * (qEnd - qBase); i += 256) sPairs[i] = a.pairs[2 * size_t(qBase) + i];
    }
    __syncthreads();

    auto product = [&](R ar, R ai, R const (&xr)[W], R const (&xi)[W], R (&yr)[W], R (&yi)[W]) __attribute__((always_inline)) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            yr[w] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, xr[w], yr[w], 0, 0, 0);
            yr[w] = __builtin_amdgcn_mfma_f64_4x4x4f64(-ai, xi[w], yr[w], 0, 0, 0);
            yi[w] = __builtin_amdgcn_mfma_f64_4x4x4f64(ar, xi[w], yi[w], 0, 0, 0);
            yi[w] = __builtin_amdgcn_mfma_f64_4x4x4f64(ai, xr[w], yi[w], 0, 0, 0);
        }
    };

    for (uint32_t it0 = 0; it0 < nItems; it0 += NS) {    // uniform over the work group
        uint32_t const it = it0 + slot;
        bool const live = (it < nItems);
        uint32_t const k = (live ? it : 0) / NSUB, y = first + k;
        size_t const off = size_t(y) * 2 * P + ex;
        EpiOps<R, EPI, W> eo;
        if (live) eo.load(a, off, P);
        R yr[W], yi[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { yr[w] = 0; yi[w] = 0; }
        if (inLds) {
            uint32_t const q0 = live ? sStarts[k] - qBase : 0, q1 = live ? sStarts[k + 1] - qBase : 0;
            for (uint32_t qb = q0; __any(qb < q1); qb += NB) {
                R ar[NB], ai[NB], xr[NB][W], xi[NB][W];
                uint32_t ia[NB], ix[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {           // (unconditional, inside the patch: all NB reads in flight at once)
                    uint32_t const qc = min(qb + u, kPairs - 1);
                    ia[u] = sPairs[2 * qc]; ix[u] = sPairs[2 * qc + 1];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    ar[u] = 0; ai[u] = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) { xr[u][w] = 0; xi[u][w] = 0; }
                    if (qb + u < q1) {
                        R const* Ab = (R const*)a.A + size_t(ia[u]) * 2 * (LM * LM);
                        R const* Xb = (R const*)a.X + size_t(ix[u]) * 2 * P;
                        ar[u] = Ab[ea]; ai[u] = Ab[LM * LM + ea]; vload<R, W>(xr[u], Xb + ex); vload<R, W>(xi[u], Xb + P + ex);
                    }
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (u > 0 && !__any(qb + u < q1)) continue;
                    product(ar[u], ai[u], xr[u], xi[u], yr, yi);
                }
            }
        } else {   // a chunk whose index data exceed the LDS patch (rows of hundreds of products): one product at a time, from global memory
            uint32_t const q0 = live ? a.starts[y] : 0, q1 = live ? a.starts[y + 1] : 0;
            for (uint32_t q = q0; __any(q < q1); ++q) {
                R ar = 0, ai = 0, xr[W], xi[W];
#pragma unroll
                for (int w = 0; w < W; ++w) { xr[w] = 0; xi[w] = 0; }
                if (q < q1) {
                    R const* Ab = (R const*)a.A + size_t(a.pairs[2 * size_t(q)]) * 2 * (LM * LM);
                    R const* Xb = (R const*)a.X + size_t(a.pairs[2 * size_t(q) + 1]) * 2 * P;
                    ar = Ab[ea]; ai = Ab[LM * LM + ea]; vload<R, W>(xr, Xb + ex); vload<R, W>(xi, Xb + P + ex);
                }
                product(ar, ai, xr, xi, yr, yi);
            }
        }
        if (live) {
            uint32_t bq = 0xffffffffu;
            if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
            epilogue_row<R, EPI, W, NPL, W>(a, off, P, yr, yi, sr, si, 0, eo, bq, ex, part, 0);
        }
    }

    if constexpr (NPL > 0) {
        // lanes that share a block column: the slots with the same column group, 4 rows each; added in a fixed order
        __shared__ double red[NPL][W][256];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int w = 0; w < W; ++w) red[p][w][t] = part[p][w];
        __syncthreads();
        for (int x = t; x < NPL * LN; x += 256) {
            int const p = x / LN, jx = x % LN, jl = (jx % CQ) / W, jw = jx % W;
            double sum = 0;
#pragma unroll 1
            for (int ss = jx / CQ; ss < NS; ss += NSUB)
                for (int r = 0; r < LM; ++r) sum += red[p][jw][(ss / 4) * 64 + r * 16 + (ss % 4) * 4 + jl];
            write_record<EPI>(a, chunk, LN, p, jx, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// ---------------------------------------------------------------------------------------------------
// 4 x 4 | 8 | 32 in float: k_spmm_small4's arithmetic (operands of a product once per thread group through a group-private LDS patch, a per-product
// sum added to the block, k = 0..3 in order: bit-identical block products) with FOUR neighbouring columns per lane instead of one.  A lane of
// k_spmm_small4 moves 4 bytes per memory instruction, 256 per wave, and these kernels are bound by the NUMBER of wave-wide memory instructions
// (profiles/r04_four_row_shapes.txt): here X and every epilogue vector move as 16-byte accesses, a thread group is 4 | 8 | 16 lanes (row i, column quad),
// a wave works on 16 | 8 | 4 block products at once.
#ifndef TFQ_S4W_NB
#define TFQ_S4W_NB 4
#endif
#ifndef TFQ_S4W_NBF
#define TFQ_S4W_NBF 4
#endif
template <int LN, int EPI, int W = 4>   // W: neighbouring columns of a lane, 4 (16-byte accesses) or 2
__global__ __launch_bounds__(256) void k_spmm_s4w(SpmmArgs a) {
    using R = float;
    if (gate_closed(a)) return;
    constexpr int LM = 4, P = LM * LN;
    constexpr int LNS = (LN > 16) ? 16 : LN;             // columns of a sub-block
    constexpr int NSUB = LN / LNS;                       // sub-blocks per block (LN = 32: 2)
    constexpr int QL = LNS / W;                          // column quads of a sub-block: 1, 2, 4
    constexpr int PE = LM * QL;                          // lanes of a thread group: 4, 8, 16
    constexpr int AV = (LM * LM) / PE;                   // elements of an A plane a lane fetches: 4, 2, 1
    constexpr int NG = 256 / PE;                         // thread groups per work group
    constexpr int NB = (EPI == EPI_NONE) ? TFQ_S4W_NB : TFQ_S4W_NBF;   // products whose operands are requested at once (the fused forms need their registers for the epilogue)
    constexpr int NPL = EpiPlanes<EPI>::N;
    static_assert(LN % LNS == 0 && LNS % W == 0 && NG % NSUB == 0, "a thread group keeps its sub-block index");
    // (the patches of the 16 | 8 | 4 groups of a wave are read by one LDS instruction: strides of 128 | 256 | 512 bytes would put them all on the
    //  same banks -- one pad of 16 bytes per column quad keeps the 16 segments of an instruction on 16 different bank quads)
    constexpr int SA = 2 * LM * LM + 4, SX = 2 * LM * LNS + 4 * ((QL * W) / 4);
    __shared__ __attribute__((aligned(16))) R AsF[NG * SA];
    __shared__ __attribute__((aligned(16))) R XsF[NG * SX];
    int const t = threadIdx.x, g = t / PE, e = t % PE;
    int const i = e / QL, jq = e % QL;
    int const j0 = (g % NSUB) * LNS + W * jq;            // first block column of this lane
    R* const As = AsF + g * SA; R* const Xs = XsF + g * SX;   // this group's patches: [re | im] planes
    uint32_t const chunk = a.order ? a.order[blockIdx.x] : blockIdx.x;   // XCD-aware launch order (tfq_plan.cpp)
    uint32_t first, last, col = 0;
    if (a.chunkFirst) { first = a.chunkFirst[chunk]; last = a.chunkFirst[chunk + 1]; col = a.chunkCol[chunk]; }
    else { first = chunk * a.CH; last = min(first + a.CH, a.nY); }

    R sr[W], si[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { sr[w] = 0; si[w] = 0; }
    if constexpr (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) {
#pragma unroll
        for (int w = 0; w < W; ++w) {
            sr[w] = ((R const*)a.sc)[(size_t(col) * 2 + 0) * LN + j0 + w];
            si[w] = ((R const*)a.sc)[(size_t(col) * 2 + 1) * LN + j0 + w];
        }
    }
    double part[NPL > 0 ? NPL : 1][W] = {};

    constexpr uint32_t kRows = 256, kPairs = 2048;       // the chunk's row ranges and index pairs in LDS, as k_spmm_small4
    __shared__ uint32_t sStarts[kRows + 1];
    __shared__ uint32_t sPairs[2 * kPairs];
    uint32_t const nRows = last - first, nItems = nRows * NSUB;   // item = sub-block of a Y block; item % NSUB == g % NSUB
    uint32_t const qBase = a.starts[first], qEnd = a.starts[last];
    bool const inLds = (nRows <= kRows) && (qEnd - qBase <= kPairs);
    if (inLds) {
        for (uint32_t x = t; x <= nRows; x += 256) sStarts[x] = a.starts[first + x];
        for (uint32_t x = t; x < 2 * (qEnd - qBase); x += 256) sPairs[x] = a.pairs[2 * size_t(qBase) + x];
    }
    __syncthreads();

    struct Ops { R a[2][AV]; R x[2][W]; };
    auto fetch = [&](Ops& o, uint32_t ia, uint32_t ix) __attribute__((always_inline)) {
        R const* Ab = (R const*)a.A + size_t(ia) * 2 * (LM * LM) + AV * e;
        R const* Xb = (R const*)a.X + size_t(ix) * 2 * P + i * LN + j0;
        vload<R, AV>(o.a[0], Ab); vload<R, AV>(o.a[1], Ab + LM * LM);
        vload<R, W>(o.x[0], Xb); vload<R, W>(o.x[1], Xb + P);
    };
    auto multiply = [&](Ops const& o, R (&yr)[W], R (&yi)[W]) __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();                 // groups never straddle a wave, LDS operations of a wave complete in order
        vstore<R, AV>(As + AV * e, o.a[0]); vstore<R, AV>(As + LM * LM + AV * e, o.a[1]);
        vstore<R, W>(Xs + i * LNS + W * jq, o.x[0]); vstore<R, W>(Xs + LM * LNS + i * LNS + W * jq, o.x[1]);
        __builtin_amdgcn_wave_barrier();
        R cr[W], ci[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { cr[w] = 0; ci[w] = 0; }
#pragma unroll
        for (int k = 0; k < LM; ++k) {
            R const ar = As[k * LM + i], ai = As[LM * LM + k * LM + i];
            R xr[W], xi[W];
            vload<R, W>(xr, Xs + k * LNS + W * jq); vload<R, W>(xi, Xs + LM * LNS + k * LNS + W * jq);
#pragma unroll
            for (int w = 0; w < W; ++w) {
                cr[w] = fma_(-ai, xi[w], fma_(ar, xr[w], cr[w]));
                ci[w] = fma_(ai, xr[w], fma_(ar, xi[w], ci[w]));
            }
        }
#pragma unroll
        for (int w = 0; w < W; ++w) { yr[w] += cr[w]; yi[w] += ci[w]; }
    };

    for (uint32_t it = g; it < nItems; it += NG) {
        uint32_t const kr = it / NSUB, y = first + kr;
        int const eb = i * LN + j0;
        size_t const off = size_t(y) * 2 * P + eb;
        EpiOps<R, EPI, W, false, (P * sizeof(R) >= 128) ? 1 : 0> eo;   // (4 x 4: a plane is 64 bytes, half a line -- no non-temporal accesses)
        eo.load(a, off, P);
        R yr[W], yi[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { yr[w] = 0; yi[w] = 0; }
        if (inLds) {
            uint32_t const q0 = sStarts[kr] - qBase, q1 = sStarts[kr + 1] - qBase;
            for (uint32_t qb = q0; qb < q1; qb += NB) {
                Ops o[NB];
                uint32_t ia[NB], ix[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {           // (unconditional, inside the patch: all NB reads in flight at once)
                    uint32_t const qc = min(qb + u, kPairs - 1);
                    ia[u] = sPairs[2 * qc]; ix[u] = sPairs[2 * qc + 1];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) if (qb + u < q1) fetch(o[u], ia[u], ix[u]);
#pragma unroll
                for (int u = 0; u < NB; ++u) if (qb + u < q1) multiply(o[u], yr, yi);
            }
        } else {   // a chunk whose index data exceed the LDS patch: one product at a time, indices from global memory
            for (uint32_t q = a.starts[y]; q < a.starts[y + 1]; ++q) {
                Ops o;
                fetch(o, a.pairs[2 * size_t(q)], a.pairs[2 * size_t(q) + 1]);
                multiply(o, yr, yi);
            }
        }
        uint32_t bq = 0xffffffffu;
        if constexpr (EPI == EPI_RESIDUAL) bq = a.bOfX ? a.bOfX[y] : y;
        epilogue_row<R, EPI, W, NPL, W>(a, off, P, yr, yi, sr, si, 0, eo, bq, eb, part, 0);
    }

    if constexpr (NPL > 0) {
        // lanes that share a block column: the groups with the same sub-block index, 4 rows each; added in a fixed order
        __shared__ double red[NPL][W][256];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int w = 0; w < W; ++w) red[p][w][t] = part[p][w];
        __syncthreads();
        for (int x = t; x < NPL * LN; x += 256) {
            int const p = x / LN, jx = x % LN, jl = (jx % LNS) / W, jw = jx % W;
            double sum = 0;
#pragma unroll 1
            for (int gg = jx / LNS; gg < NG; gg += NSUB)
                for (int r = 0; r < LM; ++r) sum += red[p][jw][gg * PE + r * QL + jl];
            write_record<EPI>(a, chunk, LN, p, jx, sum);
        }
        if (a.foldPlan) spmm_fold<R, LN, EPI>(a, col);   // small systems: the column operation behind this multiply, in the last work group of the column
    }
}

// which 4-row shapes take k_spmm_s4w
template <typename R, int LM, int LN> constexpr bool kSmall4w = (LM == 4 && sizeof(R) == 4 && LN % 4 == 0);

// which 4-row shapes take k_spmm_m4
template <typename R, int LM, int LN> constexpr bool kMfma4 = (LM == 4 && sizeof(R) == 8 && LN % 4 == 0);

// which shapes take k_spmm_mfma8: all 8-row ones; of the 4-row ones only 4 x 32 in double -- elsewhere the half-empty tile
// moves too few bytes per memory instruction and k_spmm_small4 wins (measured, 5-point stencils of 256 MB per vector,
// multiply / iteration in ms, direct | tile | small4: 4x4 z 0.74/2.44 | 0.79/2.62 | 0.33/1.73, 4x5 z 0.76/2.71 | 0.65/2.46 |
// 0.51/2.21, 4x8 z 0.71/2.35 | 0.42/1.84 | 0.34/1.65, 4x32 z 0.67/2.35 | 0.24/1.62 | 0.35/1.69, 4x4 c 0.53/2.55 | 1.39/4.25 |
// 0.45/2.41, 4x5 c 0.97/3.30 | 1.12/3.79 | 0.77/2.82, 4x8 c 0.51/1.96 | 0.72/2.53 | 0.48/2.04, 4x32 c 0.49/1.95 | 0.30/2.10 | 0.45/2.00)
template <typename R, int LM, int LN> constexpr bool kTile8 = (LM == 8) || (LM == 4 && sizeof(R) == 8 && LN == 32);

// ---------------------------------------------------------------------------------------------------
template <typename R, int LM, int LN, int EPI>
static void spmm_go(SpmmArgs const& a, uint32_t nWG, hipStream_t s) {
    if (0 == nWG) return;
    if constexpr (kSmall4w<R, LM, LN>) {   // (lab: TFQMRGPU_S4W=0 = k_spmm_small4, one column per lane; 2 = k_spmm_s4w for every launch of these shapes)
        // the multiply without epilogue gains on all three shapes (4 x 4 | 8 | 32 c: 0.387 -> 0.276, 0.267 -> 0.218, 0.265 -> 0.178 ms); the fused forms hold the
        // epilogue operands and double partial sums of four columns per lane (169 VGPRs: two waves per SIMD) and gain only where a block has many column
        // quads: 4 x 32 (-21 %); 4 x 8 is level, 4 x 4 loses 13 % (profiles/r04_four_row_shapes.txt)
        static int const use_s4w = lab_switch("TFQMRGPU_S4W", 1);
        if (use_s4w && (EPI == EPI_NONE || LN == 32 || use_s4w == 2)) { k_spmm_s4w<LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a); return; }
        // the fused launches of 4 x 8 with TWO columns per lane (129 VGPRs): 0.472 / 0.434 -> 0.419 / 0.392 ms, iteration 1.497 -> 1.398; 4 x 4 stays with k_spmm_small4 (two
        // columns per lane: 1.652 -> 1.684).  (lab: 3 = two columns per lane for both)
        if constexpr (LN < 32) if ((use_s4w == 1 && LN == 8) || use_s4w == 3) { k_spmm_s4w<LN, EPI, 2><<<dim3(nWG), dim3(256), 0, s>>>(a); return; }
    }
    if constexpr (kMfma4<R, LM, LN>) {   // (lab: TFQMRGPU_M4=0 = the kernels these shapes had before, k_spmm_small4 and the half-empty tile of k_spmm_mfma8)
        static int const use_m4 = lab_switch("TFQMRGPU_M4", 1);
        if (use_m4) { k_spmm_m4<LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a); return; }
    }
    if constexpr (LM == 16 && LN == 16 && sizeof(R) == 8) {
        if (a.ilv && a.chunkFirst) {   // the plan keeps its blocks row-pair-interleaved (tfq_plan.cpp: layoutBuffer); never the plain mode
            constexpr bool canHashI = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
            bool const hash = canHashI && a.hashV3;
            // (the first-iteration launch of EPI_XPAY_DOT is its own instance: a test of the flag per Y block costs the steady launches 0.5 %)
            constexpr bool canFirst = (EPI == EPI_XPAY_DOT);
            if (canFirst && a.first) {
                if (a.aOnce) { if (hash) k_spmm_ilv16<EPI, canHashI, true, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16<EPI, false, true, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); }
                else         { if (hash) k_spmm_ilv16<EPI, canHashI, false, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16<EPI, false, false, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); }
                return;
            }
            if (a.aOnce) { if (hash) k_spmm_ilv16<EPI, canHashI, true><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16<EPI, false, true><<<dim3(nWG), dim3(256), 0, s>>>(a); }
            else {
                // (lab: unused dynamic LDS per work group limits the work groups per CU -- 60 KiB: two, i.e. two waves per SIMD; the occupancy probe of r03)
                static size_t const padLds = size_t(lab_switch("TFQMRGPU_ILV16_LDS_KIB", 0)) << 10;
                if (hash) k_spmm_ilv16<EPI, canHashI, false><<<dim3(nWG), dim3(256), padLds, s>>>(a); else k_spmm_ilv16<EPI, false, false><<<dim3(nWG), dim3(256), 0, s>>>(a);
            }
            return;
        }
    }
    if constexpr (LM == 16 && LN == 16 && sizeof(R) == 4) {
        if (4 == a.ilv && a.chunkFirst) {   // quads of rows interleaved
            constexpr bool canHashF = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
            bool const hash = canHashF && a.hashV3;
            constexpr bool canFirst = (EPI == EPI_XPAY_DOT);
            if (canFirst && a.first) {
                if (a.aOnce) { if (hash) k_spmm_ilv16f<EPI, canHashF, true, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16f<EPI, false, true, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); }
                else         { if (hash) k_spmm_ilv16f<EPI, canHashF, false, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16f<EPI, false, false, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a); }
                return;
            }
            if (a.aOnce) { if (hash) k_spmm_ilv16f<EPI, canHashF, true><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16f<EPI, false, true><<<dim3(nWG), dim3(256), 0, s>>>(a); }
            else         { if (hash) k_spmm_ilv16f<EPI, canHashF, false><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_ilv16f<EPI, false, false><<<dim3(nWG), dim3(256), 0, s>>>(a); }
            return;
        }
    }
    if constexpr (sizeof(R) == 4 && LM % 16 == 0 && (LN == 32 || LN == 64)) {
        if (4 == a.ilv && a.chunkFirst) {
            constexpr bool canFirst = (EPI == EPI_XPAY_DOT);
            if (canFirst && a.first) k_spmm_ilvf<LM, LN, EPI, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
            else k_spmm_ilvf<LM, LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a);
            return;
        }
    }
    if constexpr (LM == 8 && LN == 8 && sizeof(R) == 8) {
        if (a.ilv && a.chunkFirst) {
            constexpr bool canHash8 = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
            constexpr bool canFirst = (EPI == EPI_XPAY_DOT);
            if (a.colBatch) {   // block columns with identical row patterns, multiplied kColBatchMax at a time (tfq_plan.cpp: colBatch; the launch runs over the first columns' chunks)
                constexpr int NBATCH = kColBatchMax;
                if (canFirst && a.first) {
                    if (canHash8 && a.hashV3) k_spmm_ilv8b<EPI, canHash8, NBATCH, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
                    else k_spmm_ilv8b<EPI, false, NBATCH, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
                    return;
                }
                if (canHash8 && a.hashV3) k_spmm_ilv8b<EPI, canHash8, NBATCH><<<dim3(nWG), dim3(256), 0, s>>>(a);
                else k_spmm_ilv8b<EPI, false, NBATCH><<<dim3(nWG), dim3(256), 0, s>>>(a);
                return;
            }
            if (canFirst && a.first) {
                if (canHash8 && a.hashV3) k_spmm_ilv8<EPI, canHash8, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
                else k_spmm_ilv8<EPI, false, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
                return;
            }
            if (canHash8 && a.hashV3) k_spmm_ilv8<EPI, canHash8><<<dim3(nWG), dim3(256), 0, s>>>(a);
            else k_spmm_ilv8<EPI, false><<<dim3(nWG), dim3(256), 0, s>>>(a);
            return;
        }
    }
    if constexpr (LM == 8 && (LN == 8 || LN == 32 || LN == 64) && sizeof(R) == 4) {
        if (4 == a.ilv && a.chunkFirst) {   // quads of rows interleaved (tfq_plan.cpp: layoutBuffer)
            constexpr bool canFirst = (EPI == EPI_XPAY_DOT);
            if (canFirst && a.first) k_spmm_ilv8f<LN, EPI, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
            else k_spmm_ilv8f<LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a);
            return;
        }
    }
    if constexpr (LM == 8 && (LN == 32 || LN == 64 || LN == 9 || LN == 10) && sizeof(R) == 8) {
        if (a.ilv && a.chunkFirst) {   // row pairs interleaved (tfq_plan.cpp: layoutBuffer)
            constexpr bool canFirst = (EPI == EPI_XPAY_DOT);
            if (canFirst && a.first) k_spmm_ilv8w<LN, EPI, canFirst><<<dim3(nWG), dim3(256), 0, s>>>(a);
            else k_spmm_ilv8w<LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a);
            return;
        }
    }
    if constexpr (LM % 16 == 0 && LN % 16 == 0) {
        // epilogue operands prefetched under the MFMAs where the registers allow it (one 16-column tile in double, two in float)
        // (not for 32 x 32 float: the prefetched operands take the fused kernels from 168 / 132 to 224 / 198 VGPRs = two waves per SIMD
        //  instead of three; measured on config 3: 0.2998 / 0.2930 ms with, 0.2947 / 0.2887 ms without)
        constexpr bool pre = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT) && ((LN / 16) * sizeof(R) <= 8) && !(sizeof(R) == 4 && LM == 32 && LN == 32);
        static int const use_pre = lab_switch("TFQMRGPU_EPI_PREFETCH", 1);
        // Three real products per complex one (Gauss) where the matrix pipe bounds the kernel: double, every shape but 16 x 16
        // (whose multiply is bound by the operand stream: 0.486 ms on P2 with either form).  Im = P3 - P1 - P2 carries the rounding
        // of the real parts: an imaginary part 10^-k times smaller than the real part loses k digits against the four-product form.
        // A drop-in caller did not ask for that, so it is OPT-IN per plan (tfqmrgpuExt_setThreeProductMultiply; until r02 it was the
        // default); never in float (the float floor of the FD fixture, 4.6e-5, moves above its threshold of 1e-4).
        // Lab builds: TFQMRGPU_3M=1 everywhere above 16 x 16, 2: 16 x 16 too.
        static int const use_m3 = lab_switch("TFQMRGPU_3M", 0);
        // the shadow vector recomputed in registers where it is the library's hash and a lane owns one column (16 x 16): the
        // fused kernels then read S/2 (`z`) or S (`c`) less (P2: 0.743 / 0.684 -> 0.719 / 0.673 ms); wider shapes and the
        // tile kernels lose and keep reading it (measured with the hash everywhere: 8 x 8 z +4 %, 32 x 32 c fused +19 %, 16 x 64 c
        // iteration +18 %, 32 x 64 c +41 %: more registers, and the hash competes with the epilogue for the vector ALU;
        // 16 x 32 z and 64 x 64 z would gain 1 %)
        constexpr bool canHash = (LN == 16) && (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
        bool const m3 = sizeof(R) == 8 && (((use_m3 || a.m3) && (LM / 16) * (LN / 16) >= 2) || use_m3 >= 2);
        bool const p = pre && use_pre;
        // unconditional (clamped) operand prefetch where a strip has at least 8 slices per block product (LM >= 32)
        constexpr bool canClamp = (LM >= 32);
        static int const use_clamp = lab_switch("TFQMRGPU_CLAMP", 1);
        // TFQMRGPU_DEEP=1 (lab builds): four operand sets for 32 x 32 float (measured on config 3: plain multiply 90.8 -> 94.8 TFLOP/s, but the
        // fused kernel with its epilogue prefetch spills: 0.317 -> 0.350 ms) -- off
        static int const use_deep = lab_switch("TFQMRGPU_DEEP", 0);
        // the stand-alone multiply on the caller's listing (plain mode), 16 x 16: a wave runs the 6-14 block products of its Y block one after the
        // other, and with two operand sets every product waits for a memory round trip -- the whole launch of BASELINE config 1's plan file
        // (1122 work groups, all resident at once) is as long as its longest chain.  More products in flight (lab: TFQMRGPU_PLAIN_NSET = 3 | 4,
        // unconditional prefetch with the index clamped): measured in profiles/r04_native_multiply.txt
        if constexpr (LM == 16 && LN == 16 && EPI == EPI_NONE) {
            static int const plainSets = lab_switch("TFQMRGPU_PLAIN_NSET", 2);
            if (!a.chunkFirst && plainSets > 2) {
                if (4 == plainSets) k_spmm_mfma<R, LM, LN, EPI, false, false, false, true, 4><<<dim3(nWG), dim3(256), 0, s>>>(a);
                else k_spmm_mfma<R, LM, LN, EPI, false, false, false, true, 3><<<dim3(nWG), dim3(256), 0, s>>>(a);
                return;
            }
        }
        auto go = [&](auto M3c, auto Hc) {
            constexpr bool M3 = decltype(M3c)::value, H = decltype(Hc)::value;
            if constexpr (canClamp) if (use_clamp) {
                constexpr int NS = (sizeof(R) == 4 && LM == 32 && LN == 32) ? 4 : 2;   // four operand sets where they fit: 32 x 32 float (a slice is 16 registers)
                if (NS == 4 && !use_deep) {
                    if (p) k_spmm_mfma<R, LM, LN, EPI, pre, M3, H, true, 2><<<dim3(nWG), dim3(256), 0, s>>>(a);
                    else k_spmm_mfma<R, LM, LN, EPI, false, M3, H, true, 2><<<dim3(nWG), dim3(256), 0, s>>>(a);
                    return;
                }
                if (p) k_spmm_mfma<R, LM, LN, EPI, pre, M3, H, true, NS><<<dim3(nWG), dim3(256), 0, s>>>(a);
                else k_spmm_mfma<R, LM, LN, EPI, false, M3, H, true, NS><<<dim3(nWG), dim3(256), 0, s>>>(a);
                return;
            }
            if (p) k_spmm_mfma<R, LM, LN, EPI, pre, M3, H><<<dim3(nWG), dim3(256), 0, s>>>(a);
            else k_spmm_mfma<R, LM, LN, EPI, false, M3, H><<<dim3(nWG), dim3(256), 0, s>>>(a);
        };
        using T = std::true_type; using F = std::false_type;
        if constexpr (canHash) {
            if (a.hashV3) { if (m3) go(T{}, T{}); else go(F{}, T{}); }
            else { if (m3) go(T{}, F{}); else go(F{}, F{}); }
        } else { if (m3) go(T{}, F{}); else go(F{}, F{}); }
    } else if constexpr (kTile8<R, LM, LN>) {
        constexpr bool pre8 = (EPI == EPI_XPAY_DOT || EPI == EPI_AXPY_NRM_DOT);
        static int const use_pre8 = lab_switch("TFQMRGPU_EPI_PREFETCH", 1);
        if (pre8 && use_pre8) k_spmm_mfma8<R, LM, LN, EPI, pre8><<<dim3(nWG), dim3(256), 0, s>>>(a);
        else k_spmm_mfma8<R, LM, LN, EPI, false><<<dim3(nWG), dim3(256), 0, s>>>(a);
    }
    else if constexpr (LM == 4) k_spmm_small4<R, LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a);
    else k_spmm_direct<R, LM, LN, EPI><<<dim3(nWG), dim3(256), 0, s>>>(a);
}

template <typename R, int LM, int LN>
static void spmm_epi(int epi, SpmmArgs const& a, uint32_t nWG, hipStream_t s) {
    switch (epi) {
    case EPI_NONE:         spmm_go<R, LM, LN, EPI_NONE>(a, nWG, s); break;
    case EPI_XPAY_DOT:     spmm_go<R, LM, LN, EPI_XPAY_DOT>(a, nWG, s); break;
    case EPI_AXPY_NRM_DOT: spmm_go<R, LM, LN, EPI_AXPY_NRM_DOT>(a, nWG, s); break;
    case EPI_RESIDUAL:     spmm_go<R, LM, LN, EPI_RESIDUAL>(a, nWG, s); break;
    }
}

#define TFQ_SIZES(X, R) \
    X(R, 4, 4) X(R, 4, 5) X(R, 4, 8) X(R, 4, 32) X(R, 8, 8) X(R, 8, 9) X(R, 8, 10) X(R, 8, 32) X(R, 8, 64) \
    X(R, 16, 16) X(R, 16, 32) X(R, 16, 64) X(R, 32, 32) X(R, 32, 64) X(R, 64, 64)

static bool spmm_dispatch(bool dbl, int lm, int ln, int epi, SpmmArgs const& a, uint32_t nWG, hipStream_t s) {
    int const key = lm * 1000 + ln;
#define TFQ_CASE(R, LM, LN) case LM * 1000 + LN: spmm_epi<R, LM, LN>(epi, a, nWG, s); return true;
    if (dbl) { switch (key) { TFQ_SIZES(TFQ_CASE, double) default: return false; } }
    else     { switch (key) { TFQ_SIZES(TFQ_CASE, float)  default: return false; } }
#undef TFQ_CASE
}

static SpmmArgs spmm_args(int epi, DevPlan const& d) {
    SpmmArgs a{};
    a.A = d.A; a.starts = d.starts; a.pairs = d.pairs; a.nY = d.nnzbX;
    a.chunkFirst = d.chunkFirst; a.chunkCol = d.chunkCol; a.CH = 0;
    a.order = d.order;
    a.ctl = d.ctl; a.v3 = d.v3; a.B = d.R ? d.R : d.B; a.bOfX = d.R ? nullptr : d.bOfX; a.pz = d.pz; a.pd = d.pd;
    a.m3 = d.m3;
    a.foldPlan = d.fold ? d.self : nullptr;
    a.hashV3 = d.hashV3; a.origCol = d.origCol; a.rowI = d.rowI; a.ilv = d.ilv; a.aOnce = d.aOnce;
    a.colBatch = d.colBatch; a.colStart = d.colStart; a.colChunkPtr = d.colChunkPtr;
    switch (epi) {
    case EPI_XPAY_DOT:     a.X = d.v6; a.Y = d.v9; a.e0 = d.v4; a.e1 = d.v8; a.sc = d.beta; a.gate = 1; a.first = d.first; break;
    case EPI_AXPY_NRM_DOT: a.X = d.v6; a.Y = d.v8; a.e0 = d.v5; a.sc = d.alfa; a.gate = 1; break;
    case EPI_RESIDUAL:     a.X = d.x;  a.Y = nullptr; a.gate = 2; break;
    default: break;
    }
    return a;
}

// (column batches: one work group per chunk of a batch's FIRST column, in an order of its own)
static bool batched(DevPlan const& d) { return d.colBatch && !d.fold; }

// the kernel family that spmm_go picks for this plan: the same conditions, in the same order
char const* spmm_kernel_family(DevPlan const& d) {
    int const LM = d.LM, LN = d.LN;
    bool const z = d.dbl;
    if (16 == LM && 16 == LN && z && d.ilv) return "k_spmm_ilv16";
    if (16 == LM && 16 == LN && !z && 4 == d.ilv) return "k_spmm_ilv16f";
    if (!z && LM % 16 == 0 && (32 == LN || 64 == LN) && 4 == d.ilv) return "k_spmm_ilvf";
    if (8 == LM && 8 == LN && z && d.ilv) return batched(d) ? "k_spmm_ilv8b" : "k_spmm_ilv8";
    if (8 == LM && (8 == LN || 32 == LN || 64 == LN) && !z && 4 == d.ilv) return "k_spmm_ilv8f";
    if (8 == LM && (32 == LN || 64 == LN || 9 == LN || 10 == LN) && z && d.ilv) return "k_spmm_ilv8w";
    if (LM % 16 == 0 && LN % 16 == 0) return "k_spmm_mfma";
    if (4 == LM && z && LN % 4 == 0 && lab_switch("TFQMRGPU_M4", 1)) return "k_spmm_m4";
    if (8 == LM || (4 == LM && z && 32 == LN)) return "k_spmm_mfma8";
    if (4 == LM && !z && (32 == LN || 8 == LN || (4 == LN && lab_switch("TFQMRGPU_S4W", 1) > 1)) && lab_switch("TFQMRGPU_S4W", 1)) return "k_spmm_s4w";   // (the fused launches; 4 x 4 c takes it without epilogue only)
    if (4 == LM) return "k_spmm_small4";
    return "k_spmm_direct";
}

void spmm_launch(int epi, DevPlan const& d, hipStream_t s) {
    if (epi != EPI_XPAY_DOT && epi != EPI_AXPY_NRM_DOT && epi != EPI_RESIDUAL) return;
    SpmmArgs a = spmm_args(epi, d);
    if (batched(d)) a.order = d.orderB; else a.colBatch = nullptr;
    spmm_dispatch(d.dbl, d.LM, d.LN, epi, a, batched(d) ? d.nChunksB : d.nChunks, s);
}

// Y = A * X on vectors of the plan (both in the plan's own block and element order), no epilogue, never gated
void spmm_apply(DevPlan const& d, void const* X, void* Y, hipStream_t s) {
    SpmmArgs a = spmm_args(EPI_NONE, d);
    a.X = X; a.Y = Y; a.gate = 0;
    if (batched(d)) a.order = d.orderB; else a.colBatch = nullptr;
    spmm_dispatch(d.dbl, d.LM, d.LN, EPI_NONE, a, batched(d) ? d.nChunksB : d.nChunks, s);
}

template <typename R, int LM, int LN>
static void epi_only(int epi, SpmmArgs const& a, uint32_t nWG, hipStream_t s) {
    if (0 == nWG) return;
    switch (epi) {
    case EPI_XPAY_DOT:     k_spmm_direct<R, LM, LN, EPI_XPAY_DOT><<<dim3(nWG), dim3(256), 0, s>>>(a); break;
    case EPI_AXPY_NRM_DOT: k_spmm_direct<R, LM, LN, EPI_AXPY_NRM_DOT><<<dim3(nWG), dim3(256), 0, s>>>(a); break;
    case EPI_RESIDUAL:     k_spmm_direct<R, LM, LN, EPI_RESIDUAL><<<dim3(nWG), dim3(256), 0, s>>>(a); break;
    }
}

void epilogue_launch(int epi, DevPlan const& d, void const* Yext, uint32_t const* i2u, hipStream_t s) {
    SpmmArgs a = spmm_args(epi, d);
    a.order = nullptr; a.Yext = Yext; a.yPerm = i2u;
    int const key = d.LM * 1000 + d.LN;
#define TFQ_CASE(R, LM, LN) case LM * 1000 + LN: epi_only<R, LM, LN>(epi, a, d.nChunks, s); break;
    if (d.dbl) { switch (key) { TFQ_SIZES(TFQ_CASE, double) default: break; } }
    else       { switch (key) { TFQ_SIZES(TFQ_CASE, float)  default: break; } }
#undef TFQ_CASE
}

// ---------------------------------------------------------------------------------------------------
// 16 x 16 blocks on the CALLER's native planes (tfqmrgpuExt_multiply: the shape of the reference's `bench multi`, whose default precision is float).  In the native order a lane
// of v_mfma_*_16x16x4 finds its operand element A[k][i] | X[k][j] 4 | 8 bytes at a time (k_spmm_mfma: 16 wave-wide loads of 256 | 512 bytes per block product, the
// memory pipe's rate -- 0.28 | 0.45 of the matrix peak on the reference's plan file).  Here a wave fetches the four planes of a product as four 16-byte-per-lane accesses
// (1 KiB each), passes them through a wave-private LDS patch and reads its operand elements from there (conflict-free: 64 consecutive floats per read).  The
// k-steps and the order of the four real products are k_spmm_mfma's: bit-identical results.  The Y block leaves through the same patch as two 1-KiB stores.
template <typename R>
__global__ __launch_bounds__(256) void k_spmm_n16(SpmmArgs a) {
    constexpr int P = 256;
    constexpr int VE = 16 / sizeof(R);               // elements of a 16-byte access: 4 | 2
    constexpr int NV = P / (64 * VE);                // accesses per lane and plane: 1 | 2
    using V = R __attribute__((ext_vector_type(VE)));
    using T4 = typename Acc<R>::T;
    int const lane = threadIdx.x & 63;
    int const wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int const lr = lane >> 4, lc = lane & 15;
    using CU32 = __attribute__((address_space(4))) uint32_t const*;
    CU32 const pairs = (CU32)(uintptr_t)a.pairs; CU32 const starts = (CU32)(uintptr_t)a.starts; CU32 const yOrder = (CU32)(uintptr_t)a.yOrder;
    __shared__ __attribute__((aligned(16))) R patch[4][4 * P];   // per wave: A re | A im | X re | X im
    R* const my = patch[wave];
    uint32_t const chunk = a.plainPer ? (blockIdx.x & 7u) * a.plainPer + (blockIdx.x >> 3) : blockIdx.x;
    uint32_t const pos = chunk * 4 + uint32_t(wave);              // (no barrier below: a wave without a Y block just leaves)
    if (pos >= a.nY) return;
    uint32_t const y = a.yOrder ? yOrder[pos] : pos;             // (a prepared order: which Y block this position computes)
    uint32_t const q0 = starts[y], q1 = starts[y + 1];
    T4 cre = T4{0, 0, 0, 0}, cim = T4{0, 0, 0, 0};
    struct Ops { V v[4][NV]; };                      // [A re | A im | X re | X im][piece]: piece n of a plane = elements 64 VE n + VE lane ...
    auto fetch = [&](Ops& o, uint32_t q) __attribute__((always_inline)) {
        R const* Ab = (R const*)a.A + size_t(pairs[2 * size_t(q)]) * 2 * P + VE * lane;
        R const* Xb = (R const*)a.X + size_t(pairs[2 * size_t(q) + 1]) * 2 * P + VE * lane;
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            o.v[0][n] = *(V const*)(Ab + 64 * VE * n); o.v[1][n] = *(V const*)(Ab + P + 64 * VE * n);
            o.v[2][n] = *(V const*)(Xb + 64 * VE * n); o.v[3][n] = *(V const*)(Xb + P + 64 * VE * n);
        }
    };
    auto mma = [&](Ops const& o) __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();             // LDS operations of one wave complete in order: the patch is free when these writes execute
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#pragma unroll
            for (int n = 0; n < NV; ++n) *(V*)(my + pl * P + 64 * VE * n + VE * lane) = o.v[pl][n];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 4; ++s) {                // k = 4 s + lr: lane (lr, lc) feeds A[k][i = lc] and X[k][j = lc]
            int const e = (4 * s + lr) * 16 + lc;
            R const ar = my[e], ai = my[P + e], xr = my[2 * P + e], xi = my[3 * P + e];
            cre = Acc<R>::mma(ar, xr, cre);
            cim = Acc<R>::mma(ar, xi, cim);
            cre = Acc<R>::mma(-ai, xi, cre);
            cim = Acc<R>::mma(ai, xr, cim);
        }
    };
    Ops o0, o1;
    if (q0 < q1) fetch(o0, q0);
    if (q0 + 1 < q1) fetch(o1, q0 + 1);
    uint32_t q = q0;
    for (; q + 2 <= q1; q += 2) {
        mma(o0);
        if (q + 2 < q1) fetch(o0, q + 2);
        mma(o1);
        if (q + 3 < q1) fetch(o1, q + 3);
    }
    if (q < q1) mma(o0);
    // the accumulator registers of a lane are rows Acc<R>::row(lane, r) of column lc: through the patch into 16 bytes per lane
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) { my[Acc<R>::row(lane, r) * 16 + lc] = cre[r]; my[P + Acc<R>::row(lane, r) * 16 + lc] = cim[r]; }
    __builtin_amdgcn_wave_barrier();
    R* const Yb = (R*)a.Y + size_t(y) * 2 * P + VE * lane;
#pragma unroll
    for (int n = 0; n < NV; ++n) {
        *(V*)(Yb + 64 * VE * n) = *(V const*)(my + 64 * VE * n + VE * lane);
        *(V*)(Yb + P + 64 * VE * n) = *(V const*)(my + P + 64 * VE * n + VE * lane);
    }
}

uint32_t multiply_blocks_per_work_group(char precision, int lm, int ln) {
    bool const dbl = ('z' == (precision | 32)) || ('d' == (precision | 32));
    bool const mfma = (lm % 16 == 0 && ln % 16 == 0);
    if (!mfma) return 0;                                   // (a prepared order is honoured by k_spmm_mfma only)
    int const mt = lm / 16;
    int const ms = (mt % 2 == 0 && 2 * (ln / 16) * (dbl ? 8 : 4) <= 32) ? 2 : 1;   // RowTiles<>::MS
    int const mu = mt / ms;
    return uint32_t((mu >= 4) ? 1 : 4 / mu);
}

tfqmrgpuStatus_t launch_multiply(char precision, int lm, int ln, uint32_t nnzbY,
    uint32_t const* starts, uint32_t const* pairs, void const* A, void const* X, void* Y, hipStream_t s, uint32_t const* yOrder)
{
    bool const dbl = ('z' == (precision | 32)) || ('d' == (precision | 32));
    SpmmArgs a{};
    a.Y = Y; a.A = A; a.X = X; a.starts = starts; a.pairs = pairs; a.nY = nnzbY;
    a.chunkFirst = nullptr; a.gate = 0;
    a.yOrder = (lm % 16 == 0 && ln % 16 == 0) ? yOrder : nullptr;
    // plain mode: enough work groups to fill 256 CUs several times, at least one strip per wave
    bool const mfma = (lm % 16 == 0 && ln % 16 == 0);
    int const mt = mfma ? lm / 16 : 1;
    int const ms = (mt % 2 == 0 && 2 * (ln / 16) * (dbl ? 8 : 4) <= 32) ? 2 : 1;   // RowTiles<>::MS
    int const mu = mt / ms;                               // strips per Y block
    uint32_t ch = (mu >= 4) ? 1 : 4 / mu;                 // one strip per wave
    if (!mfma) ch = (4 == lm) ? 64 : (lm * ln >= 256) ? 1 : 256 / (lm * ln); // k_spmm_small4: a few sub-blocks per thread group
    if (8 == lm) ch = 4;                                  // k_spmm_mfma8 (kTile8): one Y block per wave and pass
    if (4 == lm && dbl && 32 == ln) ch = 16;              // k_spmm_m4: 64 items, four per Y block (and the tile kernel's 16 before it)
    a.CH = ch;
    uint32_t nWG = (nnzbY + ch - 1) / ch;
    // (lab: contiguous eighths of the caller's Y blocks per XCD instead of round-robin work groups)
    if (mfma && nWG >= 64 && lab_switch("TFQMRGPU_PLAIN_XCD", 0)) { a.plainPer = (nWG + 7) / 8; nWG = 8 * a.plainPer; }
    // float: the reference's plan file 44.3 -> 59.2 TFLOP/s (0.28 -> 0.38 of the matrix peak), a 16 x 16 c stencil 0.360 -> 0.249 ms (0.55); double: the LDS traffic doubles with the
    // element size and the plan file LOSES 7 % (35.3 -> 32.9), P2 gains 3 %: float only (lab: TFQMRGPU_N16 bit 0 = float, bit 1 = double; profiles/r04_native_multiply.txt)
    if (16 == lm && 16 == ln && (lab_switch("TFQMRGPU_N16", 1) & (dbl ? 2 : 1))) {
        if (nWG) { if (dbl) k_spmm_n16<double><<<dim3(nWG), dim3(256), 0, s>>>(a); else k_spmm_n16<float><<<dim3(nWG), dim3(256), 0, s>>>(a); }
        return (hipSuccess == hipGetLastError()) ? TFQMRGPU_STATUS_SUCCESS : TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
    }
    if (!spmm_dispatch(dbl, lm, ln, EPI_NONE, a, nWG, s))
        return err(TFQMRGPU_BLOCKSIZE_MISSING, ln, lm);
    return (hipSuccess == hipGetLastError()) ? TFQMRGPU_STATUS_SUCCESS : TFQ_ERR(TFQMRGPU_STATUS_LAUNCH_FAILED);
}

} // namespace tfq
