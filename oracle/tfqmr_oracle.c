/*
 * TEST INFRASTRUCTURE -- not part of the product, never linked into libtfQMRgpu.so.
 * Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it.
 *
 * Plain-C restatement of the hot path of real-space/tfQMRgpu (createPlan index analysis, block
 * layout conversion, BSR multiply, AXPY/XPAY, dot/norm reductions, the three scalar updates, the
 * tfQMR driver with its stopping logic).  Every function cites the reference file:line it follows.
 *
 * Pinning: this restatement is checked (tests/test_oracle_pins.py) against
 *   - the reference itself compiled for the CPU (oracle/_ref, built by oracle/Makefile from the
 *     sources under /root/reference): bit-exact index lists, iteration counts, solution blocks;
 *   - golden vectors produced by that build and committed under tests/golden/ (the reference
 *     does not travel to the GPU box);
 *   - the reference's own known-answer tests: the Julia example's straight-line solution
 *     (example/tfqmrgpu_Julia_example.jl:117-120) and the Fortran example's A*X==B check
 *     (example/tfqmrgpu_Fortran_example.F90:108-126).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TFQO_EPS 2.5e-308 /* tfqmrgpu_linalg.hxx:31 */

typedef void (*tfqo_reduce_t)(void *ctx, double *values, int n);

#ifdef _OPENMP
#include <omp.h>
#endif
/* threads used by the multiply (the only parallel loop; the reference's solver is single threaded,
 * only its benchmark's check loop uses OpenMP, bench_tfqmrgpu.cu:358-365).  Returns the count in use. */
int tfqo_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* ---- integer analysis: restates tfqmrgpu_bsrsv_createPlan (tfqmrgpu.cu:161-339) --------------
 * Outputs (all 0-based, caller allocates):
 *   starts[nnzbX+1], pairs[2*capPairs], subset[nnzbB], colindx[nnzbX], origcol[<= nnzbX]
 * Returns the reference's status: 0, 14 (+1000*line, line arbitrary), 13 + 1000*row, 11 + 1000*count.
 * The search is the reference's linear "first match" scan (bsr.hxx:27-39), kept literally simple. */
static int32_t find_first(int32_t begin, int32_t end, int32_t value, int32_t const *array)
{
    for (int32_t q = begin; q < end; ++q) if (array[q] == value) return q;
    return -1;
}

int32_t tfqo_analyse(int32_t mb,
    int32_t const *rowPtrA, int32_t nnzbA, int32_t const *colIndA,
    int32_t const *rowPtrX, int32_t nnzbX, int32_t const *colIndX,
    int32_t const *rowPtrB, int32_t nnzbB, int32_t const *colIndB,
    int32_t off, uint64_t capPairs,
    uint32_t *starts, uint32_t *pairs, uint64_t *nPairs, uint32_t *subset,
    uint16_t *colindx, int32_t *origcol, uint32_t *nCols)
{
    if (mb < 1) return 14 + 1000 * 1;                                       /* :166 */
    if (nnzbX < 1) return 14 + 1000 * 2;                                    /* :167 */
    if (nnzbB > nnzbX) return 14 + 1000 * 3;                                /* :168 */
    if ((int64_t)nnzbA > (int64_t)mb * mb) return 14 + 1000 * 4;            /* :169 */
    if (nnzbA != rowPtrA[mb] - rowPtrA[0]) return 14 + 1000 * 5;            /* :170 */
    if (nnzbX != rowPtrX[mb] - rowPtrX[0]) return 14 + 1000 * 6;            /* :171 */
    if (nnzbB != rowPtrB[mb] - rowPtrB[0]) return 14 + 1000 * 7;            /* :172 */

    uint64_t n = 0;
    starts[0] = 0;
    for (int32_t row = 0; row < mb; ++row) {                                /* :198-219 */
        for (int32_t iy = rowPtrX[row] - off; iy < rowPtrX[row + 1] - off; ++iy) {
            int32_t const jcol = colIndX[iy];
            for (int32_t ia = rowPtrA[row] - off; ia < rowPtrA[row + 1] - off; ++ia) {
                int32_t const k = colIndA[ia] - off;
                int32_t const ix = find_first(rowPtrX[k] - off, rowPtrX[k + 1] - off, jcol, colIndX);
                if (ix >= 0) {
                    if (n >= capPairs) return -1; /* caller's buffer too small */
                    pairs[2 * n] = (uint32_t)ia; pairs[2 * n + 1] = (uint32_t)ix; ++n;
                }
            }
            starts[iy + 1] = (uint32_t)n;
        }
    }
    *nPairs = n;

    for (int32_t row = 0; row < mb; ++row) {                                /* :236-249 */
        for (int32_t ib = rowPtrB[row] - off; ib < rowPtrB[row + 1] - off; ++ib) {
            int32_t const ix = find_first(rowPtrX[row] - off, rowPtrX[row + 1] - off, colIndB[ib], colIndX);
            if (ix < 0) return 13 + 1000 * row;                             /* :245 */
            subset[ib] = (uint32_t)ix;
        }
    }

    int32_t lo = 2147483647, hi = -2147483647;                              /* :257-263 */
    for (int32_t q = 0; q < nnzbX; ++q) { if (colIndX[q] < lo) lo = colIndX[q]; if (colIndX[q] > hi) hi = colIndX[q]; }
    int64_t const nc = (int64_t)hi - lo + 1;
    if (nc < 1) return 14 + 1000 * 8;
    uint32_t *count = calloc((size_t)nc, sizeof(uint32_t));
    int32_t *jc2jb = malloc((size_t)nc * sizeof(int32_t));
    for (int32_t q = 0; q < nnzbX; ++q) ++count[colIndX[q] - lo];           /* :269-275 */
    uint32_t nb = 0;
    for (int64_t jc = 0; jc < nc; ++jc) jc2jb[jc] = count[jc] ? (int32_t)nb++ : -1; /* :280-290 */
    if (nb < 1) { free(count); free(jc2jb); return 14 + 1000 * 9; }
    for (int32_t q = 0; q < nnzbX; ++q) {                                   /* :301-310 */
        int32_t const jb = jc2jb[colIndX[q] - lo];
        origcol[jb] = colIndX[q];
        colindx[q] = (uint16_t)jb;
    }
    *nCols = nb;
    uint32_t *bcount = calloc(nb, sizeof(uint32_t));                        /* :319-336 */
    for (int32_t q = 0; q < nnzbB; ++q) ++bcount[jc2jb[colIndX[subset[q]] - lo]];
    int32_t nzero = 0;
    for (uint32_t jb = 0; jb < nb; ++jb) nzero += (bcount[jb] < 1);
    free(count); free(jc2jb); free(bcount);
    if (nzero > 0) return 11 + 1000 * nzero;                                /* :335 */
    return 0;
}

/* shadow vector of the reference CPU path: v3[i] = rand() * float(1./RAND_MAX), flat over
 * [nnzbX][2][LM][LN], glibc rand() never seeded (tfqmrgpu_linalg.hxx:799-802).  glibc's
 * generator (TYPE_3, x[i] = x[i-3] + x[i-31], seeded by 16807-Lehmer steps, first 310 outputs
 * dropped) is restated here so the values do not depend on what else the process did with rand(). */
void tfqo_shadow_glibc(float *v3, uint64_t n)
{
    uint32_t r[31];
    int32_t w = 1;
    r[0] = 1;
    for (int i = 1; i < 31; ++i) {
        int32_t const hi = w / 127773, lo = w % 127773;
        w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        r[i] = (uint32_t)w;
    }
    int f = 3, b = 0;
    float const denom = 1. / 2147483647;
    for (uint64_t i = 0; i < 310 + n; ++i) {
        r[f] += r[b];
        uint32_t const out = r[f] >> 1;
        f = (f + 1) % 31; b = (b + 1) % 31;
        if (i >= 310) v3[i - 310] = (int32_t)out * denom;
    }
}

#define REAL double
#define SUFFIX _z
#include "tfqmr_oracle_impl.h"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX _c
#include "tfqmr_oracle_impl.h"
#undef REAL
#undef SUFFIX
