// Vector kernels of the tfQMR iteration for gfx950 (MI355X): fused complex AXPY/XPAY updates,
// per-right-hand-side dot products / norms and the scalar Krylov coefficient updates.
//
// What is computed follows the reference operation by operation (real-space/tfQMRgpu
// tfqmrgpu_core.hxx:189-233 for the sequence, tfqmrgpu_linalg.hxx:629-666 col_axpay,
// :480-541 col_inner/col_reduction, :34-254 tfQMRdec35/34/T); how it is computed does not:
//  * vectors are stored column-sorted, a work group owns a "chunk" = run of blocks of ONE block
//    column, so the per-RHS scalars are loaded once per work group and reductions stay in
//    registers until one [LN] record per chunk is written (no memset + tree of launches);
//  * every lane moves 16 bytes per access (2 doubles / 4 floats) along the contiguous RHS index;
//  * updates that the reference launches one by one are fused (each vector is read/written once
//    between two scalar barriers), the second x update of an iteration rides on the next iteration;
//  * reductions are deterministic: fixed shuffle-free LDS order inside a work group, fixed chunk
//    order per column in the decision kernels.
#include <type_traits>

#include "tfq_device.hpp"
#include "tfq_vec.hpp"
#include "tfq_colops.hpp"

namespace tfq {


// ---------------------------------------------------------------------------------------------------
template <typename R> struct Vec;
template <> struct Vec<double> { static constexpr int N = 2; using T = double2; };
template <> struct Vec<float>  { static constexpr int N = 4; using T = float4;  };

// largest thread count <= 256 such that a thread's elements keep their RHS index j on every trip -- in the native order ((t * VEC) % LN == 0) and in
// the interleaved ones, where a thread's VEC elements are ROWS of one column and its column is (t + trip * T) % LN: T % LN == 0 (r04, for the 8 x 9 and
// 8 x 10 plans on row pairs; the only shape for which the two rules differ is LN = 10: 250 threads where 255 would do for the native order)
template <int LN, int VEC> constexpr int activeThreads() {
    int t = 256;
    while (t % LN) --t;
    return t;
}

template <typename R, int LM, int LN>
struct Geo {
    static constexpr int VEC = Vec<R>::N;
    static constexpr int P = LM * LN;            // elements of one plane (Re or Im) of a block
    static constexpr int IPB = P / VEC;          // vector items per block plane
    static constexpr int T = activeThreads<LN, VEC>();
    static_assert(P % VEC == 0, "block plane must be a multiple of the vector width");
};

// The vector kernels touch every element exactly once: all their accesses are non-temporal, so the streams neither
// wait for nor leave lines in the L2 that the multiply kernels want for their operand blocks (measured on P2:
// x_v6_v7 0.780 -> 0.716 ms, xpay_v6 0.316 -> 0.289 ms, and the two fused multiplies that follow 0.773/0.716 ->
// 0.743/0.681 ms).
template <typename R> __device__ inline void ldv(R (&r)[Vec<R>::N], R const* p) {
    using V = R __attribute__((ext_vector_type(Vec<R>::N)));
    auto const v = __builtin_nontemporal_load(reinterpret_cast<V const*>(p));
#pragma unroll
    for (int i = 0; i < Vec<R>::N; ++i) r[i] = v[i];
}
template <typename R> __device__ inline void stv(R* p, R const (&r)[Vec<R>::N]) {
    using V = R __attribute__((ext_vector_type(Vec<R>::N)));
    V v;
#pragma unroll
    for (int i = 0; i < Vec<R>::N; ++i) v[i] = r[i];
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
}
// the float shadow vector read with the vector width of R
template <int N> __device__ inline void ldf(float (&r)[N], float const* p) {
    if constexpr (N == 2) { auto const v = *reinterpret_cast<float2 const*>(p); r[0] = v.x; r[1] = v.y; }
    else { auto const v = *reinterpret_cast<float4 const*>(p); r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w; }
}

// per-thread copy of one complex scalar per owned RHS index
template <typename R, int LN, int VEC>
struct Scal {
    R re[VEC], im[VEC];
    // ilv = G > 0: every G consecutive elements are rows of ONE column (tfq_device.hpp: ilv_offset), else consecutive columns
    __device__ inline void load(R const* a, uint32_t col, int t, int ilv = 0) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            int const j = ilv ? ((t * VEC + v) / ilv) % LN : (t * VEC + v) % LN;
            re[v] = a[(size_t(col) * 2 + 0) * LN + j];
            im[v] = a[(size_t(col) * 2 + 1) * LN + j];
        }
    }
};

// Explicit fused multiply-adds, one fixed pattern: what the compiler contracts on its own depends on the code around the expression
// (tfq_spmm.hip has the story), and results must not move when a kernel is refactored.
__device__ inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// y := x + a*y  (tfqmrgpu_linalg.hxx:660-661)
template <typename R> __device__ inline void xpay(R& yr, R& yi, R xr, R xi, R ar, R ai) {
    R const nr = fma_(-ai, yi, fma_(ar, yr, xr));
    R const ni = fma_(ar, yi, fma_(ai, yr, xi));
    yr = nr; yi = ni;
}
// y := a*x + y  (tfqmrgpu_linalg.hxx:656-657)
template <typename R> __device__ inline void axpy(R& yr, R& yi, R xr, R xi, R ar, R ai) {
    R const nr = fma_(-ai, xi, fma_(ar, xr, yr));
    R const ni = fma_(ar, xi, fma_(ai, xr, yi));
    yr = nr; yi = ni;
}

// Sum the per-thread accumulators of a work group into out[NPL][LN] (one record per chunk).
// Fixed order => bitwise reproducible.  s is LDS of NPL*256*VEC doubles.
// The terms of one sum, in order: element m of the work group's slice belongs to column (m / G) % LN for plans that keep G rows of a
// column together (ILV = G, tfq_device.hpp: ilv_offset), to column m % LN otherwise.  (r04) ILV is a template parameter and the
// terms are fetched in batches of 16 before they are added: with the run-time `ilv` of r01-r03 every LDS read waited for the
// addition in front of it -- 32 dependent round trips by 16 threads at the end of every work group of k_v5_nrm, which is what made
// that kernel stream at 5.8 TB/s where k_xpay_v6, the same three vectors without a sum, reaches 6.6.  Same additions, same order.
template <int LN, int VEC, int T, int ILV>
__device__ inline double ordered_sum(double const* sp, int j) {
    constexpr int terms = (T * VEC) / LN, B = (terms < 16) ? terms : 16, full = terms / B * B;
    auto at = [&](int n) { return ILV ? sp[((n / ILV) * LN + j) * ILV + n % ILV] : sp[n * LN + j]; };
    double sum = 0;
    for (int n0 = 0; n0 < full; n0 += B) {
        double v[B];
#pragma unroll
        for (int i = 0; i < B; ++i) v[i] = at(n0 + i);
#pragma unroll
        for (int i = 0; i < B; ++i) sum += v[i];
    }
#pragma unroll
    for (int n = full; n < terms; ++n) sum += at(n);
    return sum;
}
template <int LN, int VEC, int T, int NPL>
__device__ inline void chunk_reduce(double (&acc)[NPL][VEC], double* s, double* out, int t, int ilv = 0, bool coherent = false) {   // coherent: folded path, tfq_colops.hpp: co_store
    __syncthreads();
    if (t < T) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int v = 0; v < VEC; ++v) s[p * (256 * VEC) + t * VEC + v] = acc[p][v];
    }
    __syncthreads();
    for (int e = t; e < NPL * LN; e += 256) {
        int const p = e / LN, j = e % LN;
        double const* sp = s + p * (256 * VEC);
        st_record(out + p * LN + j, (2 == ilv) ? ordered_sum<LN, VEC, T, 2>(sp, j) : (4 == ilv) ? ordered_sum<LN, VEC, T, 4>(sp, j) : ordered_sum<LN, VEC, T, 0>(sp, j), coherent);
    }
}

#define TFQ_CHUNK_PROLOGUE(G)                                                     \
    if (d.ctl->state != 0) return;                                                \
    int const t = threadIdx.x;                                                    \
    uint32_t const chunk = blockIdx.x;                                            \
    uint32_t const first = d.chunkFirst[chunk];                                   \
    uint32_t const nItems = (d.chunkFirst[chunk + 1] - first) * G::IPB;           \
    uint32_t const col = d.chunkCol[chunk];                                       \
    size_t const base = size_t(first) * 2 * G::P;                                 \
    (void)col;

#define TFQ_ITEM_OFFSETS(G)                                                       \
    uint32_t const blk = w / G::IPB;                                              \
    size_t const re = base + size_t(blk) * 2 * G::P + size_t(w - blk * G::IPB) * G::VEC; \
    size_t const im = re + G::P;

// At the start of a solve v5 is B scattered onto zeros (tfqmrgpu_core.hxx:153): the kernels of the first iteration that read it
// take the B block under an X block (bOfX), or zeros, directly -- v5 itself is first WRITTEN by k_v5_nrm of that iteration.
template <typename R, int VEC>
__device__ inline void ld_rhs(R (&r)[VEC], R (&i)[VEC], DevPlan const& d, uint32_t xblock, size_t inner, int P) {
    if (d.R) {   // the right-hand side of this solve is a whole X-shaped vector (mixed-precision refinement)
        R const* b = (R const*)d.R + size_t(xblock) * 2 * P + inner;
        ldv(r, b); ldv(i, b + P);
        return;
    }
    uint32_t const bq = d.bOfX[xblock];
    if (0xffffffffu == bq) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { r[v] = 0; i[v] = 0; }
    } else {
        R const* b = (R const*)d.B + size_t(bq) * 2 * P + inner;
        ldv(r, b); ldv(i, b + P);
    }
}

// ---- K0: pz <- v3 . v5 (unconjugated), start of a solve -------------------------------------------
template <typename R, int LM, int LN>
__global__ __launch_bounds__(256) void k_dot35(DevPlan d) {
    using G = Geo<R, LM, LN>;
    __shared__ double s[2 * 256 * G::VEC];
    TFQ_CHUNK_PROLOGUE(G)
    double acc[2][G::VEC] = {};
    if (t < G::T) for (uint32_t w = t; w < nItems; w += G::T) {
        TFQ_ITEM_OFFSETS(G)
        if (!d.R && 0xffffffffu == d.bOfX[first + blk]) continue;   // v5 = B scattered onto zeros: only blocks under a B block count
        R ar[G::VEC], ai[G::VEC]; float wr[G::VEC], wi[G::VEC];
        ld_rhs<R, G::VEC>(ar, ai, d, first + blk, size_t(w - blk * G::IPB) * G::VEC, G::P); ldf(wr, d.v3 + re); ldf(wi, d.v3 + im);
#pragma unroll
        for (int v = 0; v < G::VEC; ++v) {
            double const xr = ar[v], xi = ai[v], yr = wr[v], yi = wi[v];
            acc[0][v] = __builtin_fma(-xi, yi, __builtin_fma(xr, yr, acc[0][v]));
            acc[1][v] = __builtin_fma(xi, yr, __builtin_fma(xr, yi, acc[1][v]));
        }
    }
    chunk_reduce<LN, G::VEC, G::T, 2>(acc, s, d.pz + size_t(chunk) * 2 * LN, t, d.ilv);
}

// ---- KA: v6 := v5 + beta v6 ------------------------------------------------------------------------
template <typename R, int LM, int LN>
__global__ __launch_bounds__(256) void k_xpay_v6(DevPlan d) {
    using G = Geo<R, LM, LN>;
    TFQ_CHUNK_PROLOGUE(G)
    if (t >= G::T) return;
    R const* v5 = (R const*)d.v5; R* v6 = (R*)d.v6;
    Scal<R, LN, G::VEC> beta; beta.load((R const*)d.beta, col, t, d.ilv);
    auto sweep = [&](auto firstTag) __attribute__((always_inline)) {   // two loops, not a branch per trip (a branch inside cost k_x_v6_v7 2.4 %)
        for (uint32_t w = t; w < nItems; w += G::T) {
            TFQ_ITEM_OFFSETS(G)
            R xr[G::VEC], xi[G::VEC], yr[G::VEC], yi[G::VEC];
            if constexpr (decltype(firstTag)::value) {       // v5 = B on zeros, v6 = 0 at the start of a solve: not read (same arithmetic)
                ld_rhs<R, G::VEC>(xr, xi, d, first + blk, size_t(w - blk * G::IPB) * G::VEC, G::P);
#pragma unroll
                for (int v = 0; v < G::VEC; ++v) { yr[v] = 0; yi[v] = 0; }
            } else { ldv(xr, v5 + re); ldv(xi, v5 + im); ldv(yr, v6 + re); ldv(yi, v6 + im); }
#pragma unroll
            for (int v = 0; v < G::VEC; ++v) xpay(yr[v], yi[v], xr[v], xi[v], beta.re[v], beta.im[v]);
            stv(v6 + re, yr); stv(v6 + im, yi);
        }
    };
    if (d.first) sweep(std::true_type{}); else sweep(std::false_type{});
}

// ---- KC: v5 := alfa v9 + v5 ; pd <- |v5|^2 ---------------------------------------------------------
// (the v7 update that the reference does here, tfqmrgpu_core.hxx:207, needs v6 and v7 again in KD and
// is done there: same arithmetic, one read of v6 and one read+write of v7 less)
template <typename R, int LM, int LN>
__global__ __launch_bounds__(256) void k_v5_nrm(DevPlan d) {
    using G = Geo<R, LM, LN>;
    __shared__ double s[256 * G::VEC];
    TFQ_CHUNK_PROLOGUE(G)
    R* v5 = (R*)d.v5; R const* v9 = (R const*)d.v9;
    double acc[1][G::VEC] = {};
    if (t < G::T) {
        Scal<R, LN, G::VEC> alfa; alfa.load((R const*)d.alfa, col, t, d.ilv);
        auto sweep = [&](auto firstTag) __attribute__((always_inline)) {
            for (uint32_t w = t; w < nItems; w += G::T) {
                TFQ_ITEM_OFFSETS(G)
                R ar[G::VEC], ai[G::VEC], br[G::VEC], bi[G::VEC];
                ldv(ar, v9 + re); ldv(ai, v9 + im);
                if constexpr (decltype(firstTag)::value) ld_rhs<R, G::VEC>(br, bi, d, first + blk, size_t(w - blk * G::IPB) * G::VEC, G::P);   // v5 = B on zeros: first written here
                else { ldv(br, v5 + re); ldv(bi, v5 + im); }
#pragma unroll
                for (int v = 0; v < G::VEC; ++v) {
                    axpy(br[v], bi[v], ar[v], ai[v], alfa.re[v], alfa.im[v]);
                    double const r = br[v], i = bi[v];
                    acc[0][v] = __builtin_fma(i, i, __builtin_fma(r, r, acc[0][v]));
                }
                stv(v5 + re, br); stv(v5 + im, bi);
            }
        };
        if (d.first) sweep(std::true_type{}); else sweep(std::false_type{});
    }
    chunk_reduce<LN, G::VEC, G::T, 1>(acc, s, d.pd + size_t(chunk) * LN, t, d.ilv, d.fold != 0);
    if (d.fold) {   // small systems: the last work group of the column runs decT (tau, var, eta, c67) right here
        __shared__ ColScratch sc;
        fold_tail<R, LN, FOLD_DECT_C67>(d, col, sc);
    }
}

// ---- KD: [x += eta2 v7 (left over from the previous iteration)] ; v7 := v6 + c67a v7 ; x += eta v7 ;
//          v6 += alfa v4 ; v7 := v6 + c67 v7      (tfqmrgpu_core.hxx:233 | 207, 216, 218, 220)
// c67a is the coefficient of dec34, c67 the one decT has produced since; eta2 is the eta of the second
// half step of the previous iteration, whose x update is applied here unless a residual probe needed x
// earlier (ctl->xpend == 0).
template <typename R, int LM, int LN>
__global__ __launch_bounds__(256) void k_x_v6_v7(DevPlan d) {
    using G = Geo<R, LM, LN>;
    TFQ_CHUNK_PROLOGUE(G)
    if (t >= G::T) return;
    bool const pend = (d.ctl->xpend != 0);
    R* x = (R*)d.x; R* v6 = (R*)d.v6; R* v7 = (R*)d.v7; R const* v4 = (R const*)d.v4;
    Scal<R, LN, G::VEC> eta, eta2, alfa, c67, c67a;
    eta.load((R const*)d.eta, col, t, d.ilv); eta2.load((R const*)d.eta2, col, t, d.ilv); alfa.load((R const*)d.alfa, col, t, d.ilv);
    c67.load((R const*)d.c67, col, t, d.ilv); c67a.load((R const*)d.c67a, col, t, d.ilv);
    // (r04) TWO items per trip with the loads of both in front of the arithmetic: a chunk of 16 KiB is exactly two trips of the 256 threads, and with one
    //  item per trip the second trip's loads were issued behind the first trip's stores -- lab switch TFQMRGPU_VEC2, profiles/r04_vector_kernels.txt
    auto sweep = [&](auto first) __attribute__((always_inline)) {   // two loops, not a branch per trip (measured: 2.4 %)
        auto item_off = [&](uint32_t w, size_t& re, size_t& im) __attribute__((always_inline)) {
            uint32_t const blk = w / G::IPB;
            re = base + size_t(blk) * 2 * G::P + size_t(w - blk * G::IPB) * G::VEC; im = re + G::P;
        };
        for (uint32_t w0 = t; w0 < nItems; w0 += 2 * G::T) {
            R sr[2][G::VEC], si[2][G::VEC], xr[2][G::VEC], xi[2][G::VEC], ar[2][G::VEC], ai[2][G::VEC], br[2][G::VEC], bi[2][G::VEC];
            size_t re[2], im[2];
            bool const two = (w0 + G::T < nItems);
            item_off(w0, re[0], im[0]); item_off(two ? w0 + G::T : w0, re[1], im[1]);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
                if constexpr (decltype(first)::value) {          // v7 = x = 0 at the start of a solve: not read (same arithmetic on zeros)
#pragma unroll
                    for (int v = 0; v < G::VEC; ++v) { sr[u][v] = 0; si[u][v] = 0; xr[u][v] = 0; xi[u][v] = 0; }
                } else { ldv(sr[u], v7 + re[u]); ldv(si[u], v7 + im[u]); ldv(xr[u], x + re[u]); ldv(xi[u], x + im[u]); }
                ldv(ar[u], v4 + re[u]); ldv(ai[u], v4 + im[u]); ldv(br[u], v6 + re[u]); ldv(bi[u], v6 + im[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u == 1 && !two) break;
#pragma unroll
                for (int v = 0; v < G::VEC; ++v) {
                    if (pend) axpy(xr[u][v], xi[u][v], sr[u][v], si[u][v], eta2.re[v], eta2.im[v]);  // x  += eta2 v7   (previous iteration)
                    xpay(sr[u][v], si[u][v], br[u][v], bi[u][v], c67a.re[v], c67a.im[v]);           // v7  = v6 + c67a v7
                    axpy(xr[u][v], xi[u][v], sr[u][v], si[u][v], eta.re[v], eta.im[v]);             // x  += eta  v7
                    axpy(br[u][v], bi[u][v], ar[u][v], ai[u][v], alfa.re[v], alfa.im[v]);           // v6 += alfa v4
                    xpay(sr[u][v], si[u][v], br[u][v], bi[u][v], c67.re[v], c67.im[v]);             // v7  = v6 + c67 v7
                }
                stv(x + re[u], xr[u]); stv(x + im[u], xi[u]); stv(v6 + re[u], br[u]); stv(v6 + im[u], bi[u]); stv(v7 + re[u], sr[u]); stv(v7 + im[u], si[u]);
            }
        }
    };
    if (d.first) sweep(std::true_type{}); else sweep(std::false_type{});
}

// ---- KF: x := eta2 v7 + x, only when the true residual is about to be computed (tfqmrgpu_core.hxx:233) ---
template <typename R, int LM, int LN>
__global__ __launch_bounds__(256) void k_x_flush(DevPlan d) {
    using G = Geo<R, LM, LN>;
    if (d.ctl->probe == 0 || d.ctl->xpend == 0) return;
    TFQ_CHUNK_PROLOGUE(G)
    if (t >= G::T) return;
    R* x = (R*)d.x; R const* v7 = (R const*)d.v7;
    Scal<R, LN, G::VEC> eta; eta.load((R const*)d.eta2, col, t, d.ilv);
    for (uint32_t w = t; w < nItems; w += G::T) {
        TFQ_ITEM_OFFSETS(G)
        R sr[G::VEC], si[G::VEC], xr[G::VEC], xi[G::VEC];
        ldv(sr, v7 + re); ldv(si, v7 + im); ldv(xr, x + re); ldv(xi, x + im);
#pragma unroll
        for (int v = 0; v < G::VEC; ++v) axpy(xr[v], xi[v], sr[v], si[v], eta.re[v], eta.im[v]);
        stv(x + re, xr); stv(x + im, xi);
    }
}

// ---------------------------------------------------------------------------------------------------
// Column kernels: one work group per block column, several for long columns (blockIdx.y = the segment;
// the operations themselves and the segment scheme: tfq_colops.hpp)
template <typename R, int LN>
__global__ __launch_bounds__(256) void k_dec35(DevPlan d) {
    if (d.ctl->state != 0) return;
    __shared__ double s[512];
    __shared__ int last;
    col_dec35<R, LN>(d, blockIdx.x, s, int(blockIdx.y), &last);
}
template <typename R, int LN>
__global__ __launch_bounds__(256) void k_dec34(DevPlan d) {
    if (d.ctl->state != 0) return;
    __shared__ double s[512];
    __shared__ int last;
    col_dec34<R, LN>(d, blockIdx.x, s, int(blockIdx.y), &last);
}
template <typename R, int LN, bool SETC67, bool FINAL>
__global__ __launch_bounds__(256) void k_decT(DevPlan d) {
    if (d.ctl->state != 0) return;
    __shared__ double s[512];
    __shared__ double rec[2][64];
    __shared__ int last;
    col_decT<R, LN, SETC67, FINAL>(d, blockIdx.x, s, rec, int(blockIdx.y), &last);
}
template <int LN>
__global__ __launch_bounds__(256) void k_probe_col(DevPlan d) {
    if (d.ctl->state != 0 || d.ctl->probe == 0) return;
    __shared__ double s[512];
    __shared__ double rec[2][64];
    __shared__ int last;
    col_probe<LN>(d, blockIdx.x, s, rec, int(blockIdx.y), &last);
}
// single work group: max over the column records, then the stopping decision
__global__ __launch_bounds__(256) void k_decide(DevPlan d, int what, int phase) {
    if (d.ctl->state != 0) return;
    if (1 == what && 0 == d.ctl->probe) return;
    __shared__ double s0[256], s1[256];
    decide_body(d, what, phase, s0, s1);
}

// ---- start of a solve ---------------------------------------------------------------------------
// tau := |b|^2 per RHS, 1/|b|^2, rho := 1, everything else 0 (tfqmrgpu_core.hxx:121-127,154-166)
template <typename R, int LM, int LN>
__global__ __launch_bounds__(256) void k_init_col(DevPlan d, double tol, int maxIterations) {
    constexpr int P = LM * LN;
    __shared__ double s[256];
    uint32_t const col = blockIdx.x;
    int const t = threadIdx.x;
    constexpr int G = 256 / LN;
    int const g = t / LN, j = t % LN;
    double acc = 0;
    if (g < G) for (uint32_t q = d.bColPtr[col]; q < d.bColPtr[col + 1]; ++q) { // blocks of B in column order
        R const* b = (R const*)d.B + size_t(d.bList[q]) * 2 * P;
        for (int i = g; i < LM; i += G) {
            int const e = plane_offset(d.ilv, i, j, LN);
            double const r = b[e], im = b[P + e];
            acc = __builtin_fma(im, im, __builtin_fma(r, r, acc));
        }
    }
    if (g < G) s[g * LN + j] = acc;
    __syncthreads();
    if (t < LN) {
        double n2 = 0;
        for (int gg = 0; gg < G; ++gg) n2 += s[gg * LN + t];
        size_t const ir = (size_t(col) * 2 + 0) * LN + t, ii = ir + LN, i1 = size_t(col) * LN + t;
        d.tau[i1] = n2; d.invBn2[i1] = 1. / n2; d.var[i1] = 0; d.d[i1] = 0; d.status[i1] = 0;
        ((R*)d.rho)[ir] = 1; ((R*)d.rho)[ii] = 0;
        ((R*)d.alfa)[ir] = 0; ((R*)d.alfa)[ii] = 0; ((R*)d.beta)[ir] = 0; ((R*)d.beta)[ii] = 0;
        ((R*)d.c67)[ir] = 0; ((R*)d.c67)[ii] = 0; ((R*)d.eta)[ir] = 0; ((R*)d.eta)[ii] = 0;
        ((R*)d.c67a)[ir] = 0; ((R*)d.c67a)[ii] = 0; ((R*)d.eta2)[ir] = 0; ((R*)d.eta2)[ii] = 0;
    }
    if (0 == t) {   // the arrival counters of the folded column operations start every solve at zero, whatever the previous one left
        d.foldCount[col] = 0;
        if (0 == col) d.foldCount[d.nCols] = 0;
    }
    if (0 == col && 0 == t) {
        Ctl* c = d.ctl;
        double const tol2 = tol * tol;
        c->tol2 = tol2; c->target_bound2 = tol2 * 100 * 100; c->max_bound2 = 0; c->residual2_reached = 1e300; c->probe_bound2 = 0;
        for (int i = 0; i < 6; ++i) c->red[i] = 0;
        c->iteration = 0; c->maxIterations = maxIterations;
        c->state = (maxIterations > 0) ? 0 : 3;
        c->probe = 0; c->iterations_needed = maxIterations; c->nprobes = 0; c->xpend = 0; c->stallStop = 0;
    }
}

// ---- mixed precision 'm': the refinement around the float solves ---------------------------------------------------
// The reference sketches the mode as float storage with double multiply-accumulate (tfqmrgpu.cu:42, commented out) and documents it
// as "start with float and converge double" (tfqmrgpu.h:72).  Here: classical iterative refinement.  x, B and A are kept in
// double; every cycle computes r = b - A x in double, solves A d = r with the complex<float> tfQMR (all of its kernels unchanged,
// the right-hand side is the X-shaped vector R instead of B under X) and adds d to x in double.  The float iteration moves half the
// bytes of the double one and every cycle gains the digits a float solve can deliver (4-5), so the result converges to the double
// threshold at float bandwidth.

// threads t < T handle the elements w = t, t + T, ... of a chunk in the LOGICAL order [block][Re|Im][row][column]; T is a multiple of
// LN, so a thread keeps its column (right-hand side) j = t % LN
// the same for plans whose float side keeps quads of rows interleaved (16 x 16, 8 x 8 | 32 | 64, 16 | 32 x 32): an item is one 16-byte piece of R
// = rows 4 g .. 4 g + 3 of one column; on the double side that is two 16-byte row pairs (ILVZ == 2) or four elements LN apart (native order)
template <int LM, int LN, int ILVZ>
__global__ __launch_bounds__(256) void k_refine_residual_q(RefineArgs a) {
    constexpr int P = LM * LN, T = (256 / LN) * LN, Q = LM / 4, IPB = 2 * Q * LN;   // items per block
    using f4v = float __attribute__((ext_vector_type(4)));
    using d2v = double __attribute__((ext_vector_type(2)));
    __shared__ double s[256];
    DevPlan const& d = a.d;
    int const t = threadIdx.x;
    uint32_t const chunk = blockIdx.x;
    uint32_t const first = d.chunkFirst[chunk], last = d.chunkFirst[chunk + 1];
    double acc = 0;
    float* const R = (float*)d.R;
    auto quad = [&](double const* base, int g, int q, double (&v)[4]) __attribute__((always_inline)) {
        if constexpr (ILVZ == 2) {
            d2v const lo = *(d2v const*)(base + ((2 * g) * LN + q) * 2), hi = *(d2v const*)(base + ((2 * g + 1) * LN + q) * 2);
            v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = base[(4 * g + i) * LN + q];
        }
    };
    if (t < T) for (uint32_t w = t; w < (last - first) * IPB; w += T) {
        uint32_t const blk = first + w / IPB;
        int const e = int(w % IPB), c = e / (Q * LN), g = (e % (Q * LN)) / LN, q = e % LN;
        uint32_t const bq = d.bOfX[blk];
        double v[4] = {0, 0, 0, 0}, y[4] = {0, 0, 0, 0};
        if (0xffffffffu != bq) quad(a.Bz + size_t(bq) * 2 * P + size_t(c) * P, g, q, v);
        if (a.cycle > 0) quad(a.Yz + size_t(blk) * 2 * P + size_t(c) * P, g, q, y);
        f4v r;
#pragma unroll
        for (int i = 0; i < 4; ++i) { double const x = v[i] - y[i]; acc = __builtin_fma(x, x, acc); r[i] = float(x); }
        *(f4v*)(R + size_t(blk) * 2 * P + size_t(c) * P + (g * LN + q) * 4) = r;
    }
    s[t] = (t < T) ? acc : 0.;
    __syncthreads();
    if (t < LN) {
        double sum = 0;
        for (int u = t; u < T; u += LN) sum += s[u];
        d.pd[size_t(chunk) * LN + t] = sum;
    }
}

template <int LM, int LN, int ILVZ>
__global__ __launch_bounds__(256) void k_refine_update_q(RefineArgs a) {
    constexpr int P = LM * LN, Q = LM / 4, IPB = 2 * Q * LN;
    using f4v = float __attribute__((ext_vector_type(4)));
    using d2v = double __attribute__((ext_vector_type(2)));
    DevPlan const& d = a.d;
    uint32_t const chunk = blockIdx.x;
    uint32_t const first = d.chunkFirst[chunk], last = d.chunkFirst[chunk + 1];
    float const* const xc = (float const*)d.x;
    for (uint32_t w = threadIdx.x; w < (last - first) * IPB; w += 256) {
        uint32_t const blk = first + w / IPB;
        int const e = int(w % IPB), c = e / (Q * LN), g = (e % (Q * LN)) / LN, q = e % LN;
        f4v const dx = *(f4v const*)(xc + size_t(blk) * 2 * P + size_t(c) * P + (g * LN + q) * 4);
        double* const base = a.xz + size_t(blk) * 2 * P + size_t(c) * P;
        if constexpr (ILVZ == 2) {
            d2v* const lo = (d2v*)(base + ((2 * g) * LN + q) * 2); d2v* const hi = (d2v*)(base + ((2 * g + 1) * LN + q) * 2);
            d2v l = d2v{0, 0}, h = d2v{0, 0};
            if (a.cycle > 0) { l = *lo; h = *hi; }
            *lo = d2v{l[0] + double(dx[0]), l[1] + double(dx[1])}; *hi = d2v{h[0] + double(dx[2]), h[1] + double(dx[3])};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { double* const z = base + (4 * g + i) * LN + q; *z = (a.cycle > 0) ? *z + double(dx[i]) : double(dx[i]); }
        }
    }
}

template <int LM, int LN>
__global__ __launch_bounds__(256) void k_refine_residual(RefineArgs a) {
    constexpr int P = LM * LN, T = (256 / LN) * LN;
    __shared__ double s[256];
    DevPlan const& d = a.d;
    int const t = threadIdx.x;
    uint32_t const chunk = blockIdx.x;
    uint32_t const first = d.chunkFirst[chunk], last = d.chunkFirst[chunk + 1];
    double acc = 0;
    float* const R = (float*)d.R;
    if (t < T) for (uint32_t w = t; w < (last - first) * 2 * P; w += T) {
        uint32_t const blk = first + w / (2 * P);
        int const e = int(w % (2 * P)), c = e / P, r = (e % P) / LN, q = e % LN;
        size_t const zoff = size_t(c) * P + plane_offset(a.ilvZ, r, q, LN);
        uint32_t const bq = d.bOfX[blk];
        double v = (0xffffffffu == bq) ? 0. : a.Bz[size_t(bq) * 2 * P + zoff];
        if (a.cycle > 0) v -= a.Yz[size_t(blk) * 2 * P + zoff];
        acc = __builtin_fma(v, v, acc);
        R[size_t(blk) * 2 * P + size_t(c) * P + plane_offset(d.ilv, r, q, LN)] = float(v);
    }
    s[t] = (t < T) ? acc : 0.;
    __syncthreads();
    if (t < LN) {
        double sum = 0;
        for (int u = t; u < T; u += LN) sum += s[u];
        d.pd[size_t(chunk) * LN + t] = sum;
    }
}

// per block column: |r|^2 per right-hand side from the chunk records -> the set-up of the inner solve (what k_init_col does from B:
// tau = |r|^2, 1 / |r|^2, rho = 1, everything else 0) and max_rhs |r|^2 / |b|^2 for the refinement's own stopping test
template <int LN>
__global__ __launch_bounds__(256) void k_refine_init_col(RefineArgs a) {
    using R = float;
    __shared__ double s[256];
    __shared__ double rec[LN];
    DevPlan const& d = a.d;
    uint32_t const col = blockIdx.x;
    double dd[1];
    column_sum<LN, 1>(d.pd, d.colChunkPtr[col], d.colChunkPtr[col + 1], s, dd);
    int const t = threadIdx.x;
    if (t < LN) {
        double const n2 = dd[0];
        size_t const ir = (size_t(col) * 2 + 0) * LN + t, ii = ir + LN, i1 = size_t(col) * LN + t;
        if (0 == a.cycle) a.bn2z[i1] = n2;
        double const b2 = a.bn2z[i1];
        rec[t] = (b2 > 0.) ? n2 / b2 : 0.;      // a right-hand side that is zero is solved by x = 0
        d.tau[i1] = n2; d.invBn2[i1] = 1. / n2; d.var[i1] = 0; d.d[i1] = 0; d.status[i1] = 0;
        ((R*)d.rho)[ir] = 1; ((R*)d.rho)[ii] = 0;
        ((R*)d.alfa)[ir] = 0; ((R*)d.alfa)[ii] = 0; ((R*)d.beta)[ir] = 0; ((R*)d.beta)[ii] = 0;
        ((R*)d.c67)[ir] = 0; ((R*)d.c67)[ii] = 0; ((R*)d.eta)[ir] = 0; ((R*)d.eta)[ii] = 0;
        ((R*)d.c67a)[ir] = 0; ((R*)d.c67a)[ii] = 0; ((R*)d.eta2)[ir] = 0; ((R*)d.eta2)[ii] = 0;
    }
    __syncthreads();
    if (0 == t) {
        double mx = 0, bad = 0;
        for (int j = 0; j < LN; ++j) { if (rec[j] > mx) mx = rec[j]; if (!(rec[j] == rec[j]) || rec[j] > 1e300) bad = 1; }
        d.colrec[size_t(col) * 2 + 0] = mx; d.colrec[size_t(col) * 2 + 1] = bad;
    }
    if (0 == t) {
        d.foldCount[col] = 0;
        if (0 == col) d.foldCount[d.nCols] = 0;
    }
    if (0 == col && 0 == t) {
        Ctl* c = d.ctl;
        double const tol2 = a.innerTol * a.innerTol;
        c->tol2 = tol2; c->target_bound2 = tol2 * 100 * 100; c->max_bound2 = 0; c->residual2_reached = 1e300; c->probe_bound2 = 0;
        for (int i = 0; i < 6; ++i) c->red[i] = 0;
        c->iteration = 0; c->maxIterations = a.innerMaxIt;
        c->state = (a.innerMaxIt > 0) ? 0 : 3;
        c->probe = 0; c->iterations_needed = a.innerMaxIt; c->nprobes = 0; c->xpend = 0; c->stallStop = 1;
    }
}

// max over the block columns -> refine[0], refine[1] (refine[2], "a rank failed", is the host's)
__global__ __launch_bounds__(256) void k_refine_max(RefineArgs a) {
    __shared__ double s0[256], s1[256];
    int const t = threadIdx.x;
    double u = 0, v = 0;
    for (uint32_t col = t; col < a.d.nCols; col += 256) {
        double const x = a.d.colrec[size_t(col) * 2], y = a.d.colrec[size_t(col) * 2 + 1];
        if (x > u) u = x; if (y > v) v = y;
    }
    s0[t] = u; s1[t] = v;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (t < h) { if (s0[t + h] > s0[t]) s0[t] = s0[t + h]; if (s1[t + h] > s1[t]) s1[t] = s1[t + h]; }
        __syncthreads();
    }
    if (0 == t) { a.refine[0] = s0[0]; a.refine[1] = s1[0]; a.refine[2] = 0; }
}

// x (double) += d (the float solution of the inner solve); cycle 0: x = d (x is not read: nothing has to be cleared)
template <int LM, int LN>
__global__ __launch_bounds__(256) void k_refine_update(RefineArgs a) {
    constexpr int P = LM * LN;
    DevPlan const& d = a.d;
    uint32_t const chunk = blockIdx.x;
    uint32_t const first = d.chunkFirst[chunk], last = d.chunkFirst[chunk + 1];
    float const* const xc = (float const*)d.x;
    for (uint32_t w = threadIdx.x; w < (last - first) * 2 * P; w += 256) {
        uint32_t const blk = first + w / (2 * P);
        int const e = int(w % (2 * P)), c = e / P, r = (e % P) / LN, q = e % LN;
        double* const xz = a.xz + size_t(blk) * 2 * P + size_t(c) * P + plane_offset(a.ilvZ, r, q, LN);
        double const dx = double(xc[size_t(blk) * 2 * P + size_t(c) * P + plane_offset(d.ilv, r, q, LN)]);
        *xz = (a.cycle > 0) ? *xz + dx : dx;
    }
}

// ---------------------------------------------------------------------------------------------------
#define TFQ_SIZES(X, R) \
    X(R, 4, 4) X(R, 4, 5) X(R, 4, 8) X(R, 4, 32) X(R, 8, 8) X(R, 8, 9) X(R, 8, 10) X(R, 8, 32) X(R, 8, 64) \
    X(R, 16, 16) X(R, 16, 32) X(R, 16, 64) X(R, 32, 32) X(R, 32, 64) X(R, 64, 64)

template <typename R, int LM, int LN>
static hipError_t vec_run(int op, DevPlan const& d, double tol, int maxIt, hipStream_t s) {
    dim3 const grid(d.nChunks), cols(d.nCols, d.colSegMax ? d.colSegMax : 1), cols1(d.nCols), blk(256);
    switch (op) {
    case VEC_SETUP: {
        size_t const S = size_t(d.nnzbX) * 2 * LM * LN * sizeof(R);
        // nothing is cleared: x, v4, v6, v7, v8 are zero and v5 is B scattered onto zeros by definition in the first iteration, whose
        // kernels take that as given instead of reading it (DevPlan::first); v9 is written before it is read.
        // No iteration at all: x is the answer, zero.
        if (maxIt <= 0) if (auto const e = hipMemsetAsync(d.x, 0, S, s)) return e;
        if (!d.R) k_init_col<R, LM, LN><<<cols1, blk, 0, s>>>(d, tol, maxIt);   // (R: set up by k_refine_init_col, with |r|^2 from its own records)
        k_dot35<R, LM, LN><<<grid, blk, 0, s>>>(d);
    } break;
    case VEC_DEC35:    k_dec35<R, LN><<<cols, blk, 0, s>>>(d); break;
    case VEC_XPAY_V6:  k_xpay_v6<R, LM, LN><<<grid, blk, 0, s>>>(d); break;
    case VEC_DEC34:    k_dec34<R, LN><<<cols, blk, 0, s>>>(d); break;
    case VEC_V5_NRM:   k_v5_nrm<R, LM, LN><<<grid, blk, 0, s>>>(d); break;
    case VEC_DECT_C67: k_decT<R, LN, true, false><<<cols, blk, 0, s>>>(d); break;
    case VEC_X_V6_V7:  k_x_v6_v7<R, LM, LN><<<grid, blk, 0, s>>>(d); break;
    case VEC_DECT_FIN: k_decT<R, LN, false, true><<<cols, blk, 0, s>>>(d); break;
    case VEC_X_FLUSH:  k_x_flush<R, LM, LN><<<grid, blk, 0, s>>>(d); break;
    case VEC_PROBE_COL: k_probe_col<LN><<<cols, blk, 0, s>>>(d); break;
    }
    return hipSuccess;   // launch errors are sticky: run_solve collects them with hipGetLastError
}

hipError_t vec_launch(int op, DevPlan const& d, double tol, int maxIt, hipStream_t s) {
    int const key = d.LM * 1000 + d.LN;
#define TFQ_CASE(R, LM, LN) case LM * 1000 + LN: return vec_run<R, LM, LN>(op, d, tol, maxIt, s);
    if (d.dbl) { switch (key) { TFQ_SIZES(TFQ_CASE, double) default: break; } }
    else       { switch (key) { TFQ_SIZES(TFQ_CASE, float)  default: break; } }
#undef TFQ_CASE
    return hipErrorInvalidValue;
}

template <int LM, int LN>
static void refine_run(int what, RefineArgs const& a, hipStream_t s) {
    dim3 const grid(a.d.nChunks), cols(a.d.nCols), blk(256);
    bool const quads = (4 == a.d.ilv) && (0 == a.ilvZ || 2 == a.ilvZ);   // 16-byte pieces on the float side (same arithmetic, same sums)
    if (0 == what) {
        if constexpr (LM % 4 == 0) {
            if (quads && 2 == a.ilvZ) k_refine_residual_q<LM, LN, 2><<<grid, blk, 0, s>>>(a);
            else if (quads) k_refine_residual_q<LM, LN, 0><<<grid, blk, 0, s>>>(a);
            else k_refine_residual<LM, LN><<<grid, blk, 0, s>>>(a);
        } else k_refine_residual<LM, LN><<<grid, blk, 0, s>>>(a);
        k_refine_init_col<LN><<<cols, blk, 0, s>>>(a);
        k_refine_max<<<1, blk, 0, s>>>(a);
    } else {
        if constexpr (LM % 4 == 0) {
            if (quads && 2 == a.ilvZ) k_refine_update_q<LM, LN, 2><<<grid, blk, 0, s>>>(a);
            else if (quads) k_refine_update_q<LM, LN, 0><<<grid, blk, 0, s>>>(a);
            else k_refine_update<LM, LN><<<grid, blk, 0, s>>>(a);
        } else k_refine_update<LM, LN><<<grid, blk, 0, s>>>(a);
    }
}
static void refine_dispatch(int what, RefineArgs const& a, hipStream_t s) {
    int const key = a.d.LM * 1000 + a.d.LN;
#define TFQ_CASE(R, LM, LN) case LM * 1000 + LN: refine_run<LM, LN>(what, a, s); break;
    switch (key) { TFQ_SIZES(TFQ_CASE, float) default: break; }
#undef TFQ_CASE
}
void launch_refine_residual(RefineArgs const& a, hipStream_t s) { refine_dispatch(0, a, s); }
void launch_refine_update(RefineArgs const& a, hipStream_t s)   { refine_dispatch(1, a, s); }

void launch_decide(DevPlan const& d, int phase, hipStream_t s)       { k_decide<<<1, 256, 0, s>>>(d, 0, phase); }
void launch_probe_decide(DevPlan const& d, int phase, hipStream_t s) { k_decide<<<1, 256, 0, s>>>(d, 1, phase); }

} // namespace tfq
