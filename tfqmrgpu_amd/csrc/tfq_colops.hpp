// Column operations of the tfQMR iteration: the scalar Krylov coefficient updates per block column and the stopping decision, as
// DEVICE functions.  They run (a) as kernels of their own, one work group per block column (tfq_vec.hip: k_dec35 ... k_decide), and
// (b) for small systems in the TAIL of the kernel that produces their input: the work group that finishes a block column last
// (per-column arrival counter) runs the column's update, the one that finishes the last column runs the decision -- six launches
// less per iteration slot where the slot is all launch latency ("small systems").  The arithmetic and the order
// of every sum are the same in both forms: results do not depend on the form, nor on which work group happens to be last.
//
// What is computed follows the reference: tfqmrgpu_linalg.hxx:50-75 (dec35), :116-151 (dec34), :195-229 (decT),
// tfqmrgpu_core.hxx:239-260, 274-298 (the stopping logic).
#pragma once
#include "tfq_device.hpp"

namespace tfq {

#define TFQ_EPS 2.5e-308   // breakdown threshold of the reference, tfqmrgpu_linalg.hxx:31

// LDS of the column operations (one instance per kernel that calls them)
struct ColScratch { double s[512]; double rec[2][64]; double s0[256], s1[256]; int last; };

// ---------------------------------------------------------------------------------------------------
// Column operations: one work group per block column.  Sum the chunk records of the column in chunk
// order (256/LN groups of LN lanes take every G-th record, then the groups are added in order),
// then run the scalar update for the LN right-hand sides of the column.
// Coherent accesses for data that ANOTHER work group of the SAME launch has written or will read (the folded column operations, the segment shares):
// returning atomic exchanges and device-scope atomic loads.  Atomics of this part execute at the memory side (MI355X_MICROARCH.md, "Global float
// atomics"), which all eight XCDs share -- their L2s are not coherent with each other -- so neither side needs a fence: a writer waits for its
// exchanges to have RETURNED (s_waitcnt vmcnt(0): performed) in front of the arrival count, a reader uses co_load.
__device__ inline void co_store(double* p, double v) { double const old = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("" :: "v"(old)); }
__device__ inline double co_load(double const* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void st_record(double* p, double v, bool coherent) { if (coherent) co_store(p, v); else *p = v; }

// COH: the records were written by other work groups of this launch (folded path)
template <int LN, int NPL, bool COH = false>
__device__ inline void column_sum(double const* part, uint32_t c0, uint32_t c1, double* s, double (&res)[NPL]) {
    constexpr int G = 256 / LN;      // lane groups: group g takes records c0+g, c0+g+G, ...
    constexpr int U = 16;            // records in flight per lane (independent loads; the order of the sum stays fixed).  r04: every batch is predicated, so the
                                     // tail of a column is a batch too (r01-r03: 8 in flight, and up to 7 records one after the other behind the last full batch;
                                     // 24 | 12 in flight with that serial tail measured slower: P2 has 705 records per column, 44 per lane group)
    int const t = threadIdx.x, g = t / LN, j = t % LN;
    double acc[NPL] = {};
    if (g < G) {
        for (uint32_t c = c0 + g; c < c1; c += U * G) {
            double v[U][NPL];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < NPL; ++p) v[u][p] = (c + u * G < c1) ? (COH ? co_load(part + (size_t(c + u * G) * NPL + p) * LN + j) : part[(size_t(c + u * G) * NPL + p) * LN + j]) : 0.;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c + u * G < c1)
#pragma unroll
                    for (int p = 0; p < NPL; ++p) acc[p] += v[u][p];
        }
    }
    __syncthreads();
    if (g < G)
#pragma unroll
        for (int p = 0; p < NPL; ++p) s[(g * NPL + p) * LN + j] = acc[p];
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NPL; ++p) res[p] = 0;
    if (t < LN) for (int gg = 0; gg < G; ++gg)
#pragma unroll
        for (int p = 0; p < NPL; ++p) res[p] += s[(gg * NPL + p) * LN + t];
}

// ---- folding into the producer's tail (small systems) -----------------------------------------------------------------
// Arrival counter: true in every thread of the LAST work group that gets here for this counter (`expected` arrivals).  r04: no fence.  What a work
// group leaves for the last one -- chunk records, per-column stopping records -- is written with co_store, the last one reads it with co_load
// (column_sum<.., COH>, decide_body(coherent)); here every wave waits for its own exchanges to have returned, the barrier collects the waves, thread 0
// counts the work group in with a relaxed device-scope atomic.  (r01-r03: a __threadfence() per thread and two more in thread 0 -- every one an L2
// write-back -- which is why folding LOST from 256 chunks on; the crossover with this form: profiles/r04_small_systems.txt.)
__device__ inline bool fold_arrive(uint32_t* counter, uint32_t expected, int* lastFlag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's record exchanges (and whatever else it had in flight) are performed
    __syncthreads();
    if (0 == threadIdx.x) {
        uint32_t const before = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int const last = (before + 1 == expected) ? 1 : 0;
        if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next use
        *lastFlag = last;
    }
    __syncthreads();
    return *lastFlag != 0;
}

// ---- columns of many chunks: several work groups per column (r04) ------------------------------------------------------------
// One work group per block column is 2-49 work groups on a 256-CU part: a column of BASELINE config 3, 4 or 5 has 4096 chunk records, and
// k_dec35 / k_dec34 / k_decT took 17-35 us each there -- 9 % of config 3's iteration, 4 % of config 4's and 5's -- in a kernel that is nothing
// but a few round trips to memory per lane group.  Long columns (col_segments below) are summed by S work groups:
// work group `seg` sums its share of the records (contiguous, balanced: the same lane-group order as column_sum inside the share), leaves it
// in DevPlan::colPart, and the LAST one to arrive (the column's arrival counter, fold_arrive) adds the S shares IN ORDER and runs the
// column's update.  The order of every sum is fixed by (n, LN) alone: results do not depend on which work group is last, nor on the
// number of ranks (a column's chunks are the same on any rank).  Short columns -- every golden fixture -- are summed by one work group exactly
// as before.  seg < 0: one work group for the whole column whatever its length (the folded path).
// A segment is what one work group sums in ONE round of loads (256 / LN lane groups x 16 records in flight: 256 records for LN = 16, 128 for 32); columns
// of at most four such rounds stay with one work group -- the arrival costs a few microseconds (P2's columns of 705 records, three rounds: 11 us per
// column kernel either way), BASELINE config 3's, 4's and 5's columns of 1024-4096 records are cut (17-37 -> 10-17 us, profiles/r04_vector_kernels.txt).
__host__ __device__ inline uint32_t col_seg_len(int LN) { return uint32_t(256 / LN) * 16u; }
__host__ __device__ inline uint32_t col_segments(uint32_t n, int LN) { return (n <= 4 * col_seg_len(LN)) ? 1u : (n + col_seg_len(LN) - 1) / col_seg_len(LN); }
constexpr uint32_t kColSlot = 64;      // granule of the slot numbering below (<= the shortest segment: LN = 64)
// the slot of (column, segment) in colPart: c0 / K + col + seg is unique for any K <= the segment length (floor(c1 / K) + 1 >= floor(c0 / K) + ceil(n / K) >= segments)
__device__ inline size_t col_part_slot(uint32_t c0, uint32_t col, uint32_t seg) { return size_t(c0 / kColSlot) + col + seg; }

template <int LN, int NPL>
__device__ inline bool column_total(DevPlan const& d, double const* part, uint32_t col, int seg, double* s, int* lastFlag, double (&res)[NPL]) {
    uint32_t const c0 = d.colChunkPtr[col], c1 = d.colChunkPtr[col + 1], n = c1 - c0;
    uint32_t const S = (seg < 0) ? 1u : col_segments(n, LN);
    if (uint32_t(seg < 0 ? 0 : seg) >= S) return false;          // (the grid has as many segments as the longest column needs)
    if (1 == S) { if (seg < 0) column_sum<LN, NPL, true>(part, c0, c1, s, res); else column_sum<LN, NPL>(part, c0, c1, s, res); return true; }
    uint32_t const len = (n + S - 1) / S, b0 = c0 + uint32_t(seg) * len, b1 = min(c1, b0 + len);
    column_sum<LN, NPL>(part, b0, b1, s, res);
    int const t = threadIdx.x;
    // The shares travel as RETURNING atomic exchanges and are read back as device-scope atomic loads: atomics of this part execute at the memory side
    // (MI355X_MICROARCH.md, "Global float atomics"), which is coherent for all eight XCDs -- no release / acquire FENCE anywhere.  (The first form of this
    // scheme used fold_arrive: every thread's __threadfence() is an L2 write-back, and with 32 columns x 16 segments = 512 work groups per launch the column
    // kernels of config 4's shard went from 37 to 77 us; profiles/r04_vector_kernels.txt.)  Only wave 0 writes (LN <= 64 lanes), and thread 0 of that wave
    // counts the work group in: one s_waitcnt for the wave's exchanges to have returned -- i.e. to have been performed -- orders them in front of the count.
    if (t < LN)
#pragma unroll
        for (int p = 0; p < NPL; ++p) co_store(d.colPart + (col_part_slot(c0, col, uint32_t(seg)) * 3 + p) * LN + t, res[p]);
    if (t < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (0 == t) {
        uint32_t const before = __hip_atomic_fetch_add(d.foldCount + col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int const last = (before + 1 == S) ? 1 : 0;
        if (last) __hip_atomic_store(d.foldCount + col, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next use
        *lastFlag = last;
    }
    __syncthreads();
    if (0 == *lastFlag) return false;
    if (t < LN) {
#pragma unroll
        for (int p = 0; p < NPL; ++p) res[p] = 0;
        constexpr int U = 8;
        for (uint32_t g0 = 0; g0 < S; g0 += U) {                  // the shares in order, U loads in flight
            double v[U][NPL];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int p = 0; p < NPL; ++p)
                    v[u][p] = (g0 + u < S) ? co_load(d.colPart + (col_part_slot(c0, col, g0 + u) * 3 + p) * LN + t) : 0.;
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (g0 + u < S)
#pragma unroll
                    for (int p = 0; p < NPL; ++p) res[p] += v[u][p];
        }
    }
    return true;
}

// dec35: beta = z/rho, rho = z  (tfqmrgpu_linalg.hxx:50-75)
template <typename R, int LN>
__device__ inline void col_dec35(DevPlan const& d, uint32_t const col, double* s, int seg, int* lastFlag) {
    // (r04: the per-RHS scalars of the update are requested in FRONT of the column's sum -- they do not depend on it, and behind it each of them was one more
    //  round trip to memory in a kernel that is nothing but round trips; lanes beyond LN read the last right-hand side's, harmlessly)
    int const j = threadIdx.x, jc = (j < LN) ? j : LN - 1;
    size_t const ir = (size_t(col) * 2 + 0) * LN + jc, ii = ir + LN;
    R* rho = (R*)d.rho; R* bet = (R*)d.beta;
    double const rr = double(rho[ir]), ri = double(rho[ii]);
    double z[2];
    if (!column_total<LN, 2>(d, d.pz, col, seg, s, lastFlag, z)) return;
    if (j >= LN) return;
    double const abs2rho = rr * rr + ri * ri, abs2z = z[0] * z[0] + z[1] * z[1];
    d.z[ir] = z[0]; d.z[ii] = z[1];
    if (abs2z < TFQ_EPS || abs2rho < TFQ_EPS) {
        d.status[size_t(col) * LN + j] = -1;
        bet[ir] = 0; bet[ii] = 0; rho[ir] = 0; rho[ii] = 0;
    } else {
        double const den = 1. / abs2rho;
        bet[ir] = R((z[0] * rr + z[1] * ri) * den);
        bet[ii] = R((z[1] * rr - z[0] * ri) * den);
        rho[ir] = R(z[0]); rho[ii] = R(z[1]);
    }
}

// dec34: alfa = -rho/z, c67 = z*(var*eta/rho)  (tfqmrgpu_linalg.hxx:116-151)
template <typename R, int LN>
__device__ inline void col_dec34(DevPlan const& d, uint32_t const col, double* s, int seg, int* lastFlag) {
    int const j = threadIdx.x, jc = (j < LN) ? j : LN - 1;
    size_t const ir = (size_t(col) * 2 + 0) * LN + jc, ii = ir + LN;
    // the latest eta is the one of the second half step of the previous iteration (eta2)
    R const* rho = (R const*)d.rho; R const* eta = (R const*)d.eta2; R* alf = (R*)d.alfa; R* c67 = (R*)d.c67a;
    double const rr = double(rho[ir]), ri = double(rho[ii]);                     // (requested in front of the column's sum, like everything it does not depend on)
    double const er = double(eta[ir]), ei = double(eta[ii]), var0 = d.var[size_t(col) * LN + jc];
    double z[2];
    if (!column_total<LN, 2>(d, d.pz, col, seg, s, lastFlag, z)) return;
    if (j >= LN) return;
    double const abs2rho = rr * rr + ri * ri, abs2z = z[0] * z[0] + z[1] * z[1];
    d.z[ir] = z[0]; d.z[ii] = z[1];
    if (abs2z < TFQ_EPS || abs2rho < TFQ_EPS) {
        d.status[size_t(col) * LN + j] = -2;
        alf[ir] = 0; alf[ii] = 0; c67[ir] = 0; c67[ii] = 0;
    } else {
        double const zden = -1. / abs2z;
        alf[ir] = R((rr * z[0] + ri * z[1]) * zden);
        alf[ii] = R((ri * z[0] - rr * z[1]) * zden);
        double const vden = var0 / abs2rho;
        double const tr = (er * rr + ei * ri) * vden, ti = (ei * rr - er * ri) * vden;
        c67[ir] = R(z[0] * tr - z[1] * ti);
        c67[ii] = R(z[1] * tr + z[0] * ti);
    }
}

// decT: var = d/tau, c = 1/(1+var), tau = d c, eta = -c alfa [, c67 = var c]  (tfqmrgpu_linalg.hxx:195-229)
// FINAL additionally leaves the per-column record for the stopping test:
//   colrec[col] = { max_j tau_j/|b_j|^2 , 1 if any RHS of the column is not broken down (-1/-2) }
template <typename R, int LN, bool SETC67, bool FINAL>
__device__ inline void col_decT(DevPlan const& d, uint32_t const col, double* s, double (*rec)[64], int seg, int* lastFlag) {
    int const j = threadIdx.x, jc = (j < LN) ? j : LN - 1;
    double const Tau0 = d.tau[size_t(col) * LN + jc];                           // (requested in front of the column's sum, like everything it does not depend on)
    int8_t const st0 = d.status[size_t(col) * LN + jc];
    R const alr0 = ((R const*)d.alfa)[(size_t(col) * 2 + 0) * LN + jc], ali0 = ((R const*)d.alfa)[(size_t(col) * 2 + 1) * LN + jc];
    double const invB0 = FINAL ? d.invBn2[size_t(col) * LN + jc] : 0.;
    double dd[1];
    if (!column_total<LN, 1>(d, d.pd, col, seg, s, lastFlag, dd)) return;
    if (j < LN) {
        size_t const ir = (size_t(col) * 2 + 0) * LN + j, ii = ir + LN, i1 = size_t(col) * LN + j;
        R* eta = (R*)(FINAL ? d.eta2 : d.eta); R* c67 = (R*)d.c67;
        double cosi = 0; R r67 = 1;
        double const Tau = Tau0;
        int8_t st = st0;
        double newTau;
        if (fabs(Tau) > TFQ_EPS) {
            double const Var = dd[0] / Tau;
            cosi = 1. / (1. + Var);
            d.var[i1] = Var;
            newTau = dd[0] * cosi;
            r67 = R(Var * cosi);
        } else {
            st = -3; d.status[i1] = -3;
            d.var[i1] = 0; newTau = 0;
        }
        d.tau[i1] = newTau;
        d.d[i1] = dd[0];
        if (st < 0) { eta[ir] = 0; eta[ii] = 0; }
        else { eta[ir] = R(-cosi * alr0); eta[ii] = R(-cosi * ali0); }
        if (SETC67) { c67[ir] = r67; c67[ii] = 0; }
        if (FINAL) { rec[0][j] = newTau * invB0; rec[1][j] = (st == -1 || st == -2) ? 0. : 1.; }
    }
    if (FINAL) {
        __syncthreads();
        if (0 == j) {
            double mx = 0, alive = 0; // max ignores NaN like std::max(a, nan) == a in the reference loop
            for (int jj = 0; jj < LN; ++jj) { if (rec[0][jj] > mx) mx = rec[0][jj]; if (rec[1][jj] > alive) alive = rec[1][jj]; }
            st_record(d.colrec + size_t(col) * 2 + 0, mx, seg < 0); st_record(d.colrec + size_t(col) * 2 + 1, alive, seg < 0);   // (folded: read by another work group of this launch)
        }
    }
}

// true residual per RHS of one column after the probe multiply (tfqmrgpu_core.hxx:274-286):
//   colrec[col] = { max_j res2_j , 1 if any RHS with status 0 has res2 > tol2 }
template <int LN>
__device__ inline void col_probe(DevPlan const& d, uint32_t const col, double* s, double (*rec)[64], int seg, int* lastFlag) {
    double dd[1];
    if (!column_total<LN, 1>(d, d.pd, col, seg, s, lastFlag, dd)) return;
    int const j = threadIdx.x;
    if (j < LN) {
        size_t const i1 = size_t(col) * LN + j;
        double const res2 = dd[0] * d.invBn2[i1];
        double const tol2 = d.ctl->tol2;
        double open = 0;
        if (res2 > tol2) { if (0 == d.status[i1]) open = 1; }
        else if (res2 <= 0) d.status[i1] = 1;
        rec[0][j] = res2; rec[1][j] = open;
    }
    __syncthreads();
    if (0 == j) {
        double mx = 0, open = 0;
        for (int jj = 0; jj < LN; ++jj) { if (rec[0][jj] > mx) mx = rec[0][jj]; if (rec[1][jj] > open) open = rec[1][jj]; }
        st_record(d.colrec + size_t(col) * 2 + 0, mx, seg < 0); st_record(d.colrec + size_t(col) * 2 + 1, open, seg < 0);   // (folded: the deciding work group is another one of this launch)
    }
}

// ---- single work group: max over the column records, then the stopping decision -------------------
// what = 0: end of an iteration (tfqmrgpu_core.hxx:239-260), 1: after a probe (:287-298)
// phase 0: reduce + decide, 1: reduce only into ctl->red (an all-reduce follows), 2: decide only
__device__ inline void decide_body(DevPlan const& d, int what, int phase, double* s0, double* s1, bool coherent = false) {   // coherent: the column records come from other work groups of THIS launch (folded path)
    Ctl* c = d.ctl;
    int const t = threadIdx.x;
    if (phase != 2) {
        double a = 0, b = 0;
        for (uint32_t col = t; col < d.nCols; col += 256) {
            double const u = coherent ? co_load(d.colrec + size_t(col) * 2) : d.colrec[size_t(col) * 2], v = coherent ? co_load(d.colrec + size_t(col) * 2 + 1) : d.colrec[size_t(col) * 2 + 1];
            if (u > a) a = u; if (v > b) b = v;
        }
        s0[t] = a; s1[t] = b;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if (t < h) { if (s0[t + h] > s0[t]) s0[t] = s0[t + h]; if (s1[t + h] > s1[t]) s1[t] = s1[t + h]; }
            __syncthreads();
        }
        if (0 == t) { c->red[3 * what] = s0[0]; c->red[3 * what + 1] = s1[0]; }   // red[3 * what + 2] is the host's: "this rank failed"
        __syncthreads();
    }
    if (phase == 1 || t != 0) return;
    if (c->red[3 * what + 2] > 0.) { c->state = 4; c->probe = 0; return; }   // some rank failed: every rank stops at this very slot
    if (0 == what) {
        int const it = ++c->iteration;
        double const bound2 = c->red[0] * (2 * it + 1);
        c->max_bound2 = bound2;
        int probe = (bound2 <= c->target_bound2) || (it >= c->maxIterations);
        if (c->red[1] == 0.) { c->state = 2; probe = 0; } // every right-hand side broke down
        c->probe = probe;
        c->xpend = 1;   // x += eta2 v7 of this iteration is still outstanding
    } else {
        double max_res2 = 1.4e-76;
        if (c->red[3] > max_res2) max_res2 = c->red[3];
        c->target_bound2 = (c->max_bound2 / max_res2) * c->tol2;
        c->nprobes += 1;
        c->probe = 0;
        c->xpend = 0;   // k_x_flush has brought x up to date
        double const before = c->residual2_reached;      // of the previous probe of this solve (1e300 at the start)
        double const promised = (c->probe_bound2 > 0.) ? c->max_bound2 / c->probe_bound2 : 0.;   // what the bound has gained since then (squared)
        c->residual2_reached = max_res2;
        c->probe_bound2 = c->max_bound2;
        if (c->red[4] == 0.) { c->iterations_needed = c->iteration; c->state = 1; }
        else if (c->iteration >= c->maxIterations) c->state = 3;
        // the float floor (inner solves of 'm'): |r| no better than 0.7 x the previous probe's, or -- seen one probe earlier -- |r| improved 3 x less
        // than the recurrence's bound did while gaining less than a digit: the recurrence has come loose from the true residual
        else if (c->stallStop && (max_res2 > 0.49 * before || (promised > 0. && max_res2 > 9. * promised * before && max_res2 > 0.01 * before))) c->state = 3;
    }
}


enum { FOLD_DEC34 = 1, FOLD_DECT_C67 = 2, FOLD_DECT_FINAL = 3, FOLD_PROBE = 4 };

// called by every thread of a work group that has just written the chunk record(s) of block column `col`
template <typename R, int LN, int WHAT>
__device__ inline void fold_tail(DevPlan const& d, uint32_t col, ColScratch& sc) {
    uint32_t const chunksOfCol = d.colChunkPtr[col + 1] - d.colChunkPtr[col];
    if (!fold_arrive(d.foldCount + col, chunksOfCol, &sc.last)) return;
    if constexpr (WHAT == FOLD_DEC34) col_dec34<R, LN>(d, col, sc.s, -1, &sc.last);
    else if constexpr (WHAT == FOLD_DECT_C67) col_decT<R, LN, true, false>(d, col, sc.s, sc.rec, -1, &sc.last);
    else if constexpr (WHAT == FOLD_DECT_FINAL) col_decT<R, LN, false, true>(d, col, sc.s, sc.rec, -1, &sc.last);
    else col_probe<LN>(d, col, sc.s, sc.rec, -1, &sc.last);
    if constexpr (WHAT == FOLD_DECT_FINAL || WHAT == FOLD_PROBE) {     // the column's record for the decision is written: last column decides
        if (!fold_arrive(d.foldCount + d.nCols, d.nCols, &sc.last)) return;
        decide_body(d, (WHAT == FOLD_PROBE) ? 1 : 0, 0, sc.s0, sc.s1, true);
    }
}

} // namespace tfq
