/*
 * TEST INFRASTRUCTURE -- not part of the product.  Included twice by tfqmr_oracle.c, once with
 * REAL = double (suffix _z) and once with REAL = float (suffix _c).
 *
 * Floating-point half of the CPU restatement of real-space/tfQMRgpu's hot path.  All arrays are in
 * the caller's BSR block order and in the reference's device-native element order
 *     v[nnzb][2][LM][LN]   (Re plane, then Im plane; the RHS index is contiguous)
 *     A[nnzbA][2][LM(k)][LM(i)]  = blocks of A stored TRANSPOSED, as the reference's CUDA path and its
 *                                  manual define them (tfqmrgpu.cu:514-517, tfqmrgpu_blockmult.hxx:54).
 * The reference's own HAS_NO_CUDA multiply reads A un-transposed (tfqmrgpu_blocksparse.hxx:167-168,
 * marked "ToDo" there); this file implements the documented semantics, so comparisons with the
 * compiled reference CPU library pass the A flag flipped (see oracle/README.md).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* Y = A*X over the pair list: restates gemmNxNf (tfqmrgpu_blockmult.hxx:29-92) and the host loop
 * around it (tfqmrgpu_blocksparse.hxx:139-193): per Y block, pairs in list order, per pair a full
 * k-contraction that is then added to the block accumulator. */
void FN(tfqo_spmm)(int LM, int LN, uint32_t nnzbY, uint32_t const *starts, uint32_t const *pairs,
                   REAL const *A, REAL const *X, REAL *Y)
{
    size_t const P = (size_t)LM * LN, Q = (size_t)LM * LM;
    long iy;
#pragma omp parallel for schedule(dynamic, 16)
    for (iy = 0; iy < (long)nnzbY; ++iy) {
        REAL *y = Y + (size_t)iy * 2 * P;
        for (size_t e = 0; e < 2 * P; ++e) y[e] = 0;
        for (uint32_t p = starts[iy]; p < starts[iy + 1]; ++p) {
            REAL const *a = A + (size_t)pairs[2 * (size_t)p] * 2 * Q;
            REAL const *x = X + (size_t)pairs[2 * (size_t)p + 1] * 2 * P;
            for (int i = 0; i < LM; ++i) {
                for (int j = 0; j < LN; ++j) {
#ifdef TFQO_RUNNING_SUM   /* sensitivity probe only (tests/test_oracle_sensitivity.py): ONE running sum over all products of a Y element */
                    REAL re = y[(size_t)i * LN + j], im = y[P + (size_t)i * LN + j];
#else
                    REAL re = 0, im = 0;
#endif
                    for (int k = 0; k < LM; ++k) {
                        REAL const ar = a[(size_t)k * LM + i], ai = a[Q + (size_t)k * LM + i];
                        REAL const xr = x[(size_t)k * LN + j], xi = x[P + (size_t)k * LN + j];
                        re += ar * xr - ai * xi;
                        im += ar * xi + ai * xr;
                    }
#ifdef TFQO_RUNNING_SUM
                    y[(size_t)i * LN + j] = re;
                    y[P + (size_t)i * LN + j] = im;
#else
                    y[(size_t)i * LN + j] += re;
                    y[P + (size_t)i * LN + j] += im;
#endif
                }
            }
        }
    }
}

/* y := a*x + y with one complex scalar per (block column, RHS) -- col_axpay<ScaleX=true>,
 * tfqmrgpu_linalg.hxx:646-657 */
void FN(tfqo_axpy)(int LM, int LN, uint32_t nnzb, uint16_t const *colindx, REAL *y, REAL const *x, REAL const *a)
{
    size_t const P = (size_t)LM * LN;
    for (uint32_t b = 0; b < nnzb; ++b) {
        REAL const *s = a + (size_t)colindx[b] * 2 * LN;
        REAL *yb = y + (size_t)b * 2 * P; REAL const *xb = x + (size_t)b * 2 * P;
        for (int i = 0; i < LM; ++i) for (int j = 0; j < LN; ++j) {
            size_t const e = (size_t)i * LN + j;
            REAL const xr = xb[e], xi = xb[P + e], yr = yb[e], yi = yb[P + e];
            yb[e]     = s[j] * xr - s[LN + j] * xi + yr;
            yb[P + e] = s[LN + j] * xr + s[j] * xi + yi;
        }
    }
}

/* y := x + a*y -- col_axpay<ScaleX=false>, tfqmrgpu_linalg.hxx:658-662 */
void FN(tfqo_xpay)(int LM, int LN, uint32_t nnzb, uint16_t const *colindx, REAL *y, REAL const *a, REAL const *x)
{
    size_t const P = (size_t)LM * LN;
    for (uint32_t b = 0; b < nnzb; ++b) {
        REAL const *s = a + (size_t)colindx[b] * 2 * LN;
        REAL *yb = y + (size_t)b * 2 * P; REAL const *xb = x + (size_t)b * 2 * P;
        for (int i = 0; i < LM; ++i) for (int j = 0; j < LN; ++j) {
            size_t const e = (size_t)i * LN + j;
            REAL const xr = xb[e], xi = xb[P + e], yr = yb[e], yi = yb[P + e];
            yb[e]     = xr + s[j] * yr - s[LN + j] * yi;
            yb[P + e] = xi + s[LN + j] * yr + s[j] * yi;
        }
    }
}

/* d[col][0|1][j] = sum over blocks of the column, sum_k x*w (no conjugation), w is the float shadow
 * vector; operands promoted to double before the product -- host branch of dotp,
 * tfqmrgpu_linalg.hxx:566-585 */
void FN(tfqo_dotp)(int LM, int LN, uint32_t nnzb, uint32_t nCols, uint16_t const *colindx,
                   double *d, REAL const *x, float const *w)
{
    size_t const P = (size_t)LM * LN;
    for (size_t e = 0; e < (size_t)nCols * 2 * LN; ++e) d[e] = 0;
    for (uint32_t b = 0; b < nnzb; ++b) {
        double *dc = d + (size_t)colindx[b] * 2 * LN;
        REAL const *xb = x + (size_t)b * 2 * P; float const *wb = w + (size_t)b * 2 * P;
        for (int j = 0; j < LN; ++j) {
            double re = 0, im = 0;
            for (int k = 0; k < LM; ++k) {
                double const xr = xb[(size_t)k * LN + j], xi = xb[P + (size_t)k * LN + j];
                double const wr = wb[(size_t)k * LN + j], wi = wb[P + (size_t)k * LN + j];
                re += xr * wr - xi * wi;
                im += xr * wi + xi * wr;
            }
            dc[j] += re; dc[LN + j] += im;
        }
    }
}

/* d[col][j] = sum |x|^2 -- host branch of nrm2, tfqmrgpu_linalg.hxx:610-623 */
void FN(tfqo_nrm2)(int LM, int LN, uint32_t nnzb, uint32_t nCols, uint16_t const *colindx, double *d, REAL const *x)
{
    size_t const P = (size_t)LM * LN;
    for (size_t e = 0; e < (size_t)nCols * LN; ++e) d[e] = 0;
    for (uint32_t b = 0; b < nnzb; ++b) {
        double *dc = d + (size_t)colindx[b] * LN;
        REAL const *xb = x + (size_t)b * 2 * P;
        for (int j = 0; j < LN; ++j) {
            double s = 0;
            for (int k = 0; k < LM; ++k) {
                double const xr = xb[(size_t)k * LN + j], xi = xb[P + (size_t)k * LN + j];
                s += xr * xr + xi * xi;
            }
            dc[j] += s;
        }
    }
}

/* v[subset[b]] += scal * B[b] -- add_RHS, tfqmrgpu_linalg.hxx:421-426 */
void FN(tfqo_add_rhs)(int LM, int LN, uint32_t nnzbB, uint32_t const *subset, REAL *v, REAL const *B, REAL scal)
{
    size_t const E = (size_t)2 * LM * LN;
    for (uint32_t b = 0; b < nnzbB; ++b)
        for (size_t e = 0; e < E; ++e) v[(size_t)subset[b] * E + e] += scal * B[(size_t)b * E + e];
}

/* beta = z/rho, rho = z; breakdown -> status -1 -- tfQMRdec35, tfqmrgpu_linalg.hxx:50-75 */
void FN(tfqo_dec35)(int LN, uint32_t nCols, int8_t *status, REAL *rho, REAL *beta, double const *z)
{
    for (uint32_t c = 0; c < nCols; ++c) for (int j = 0; j < LN; ++j) {
        size_t const r = ((size_t)c * 2) * LN + j, i = r + LN;
        double const rr = rho[r], ri = rho[i], zr = z[r], zi = z[i];
        double const n2rho = rr * rr + ri * ri, n2z = zr * zr + zi * zi;
        if (n2z < TFQO_EPS || n2rho < TFQO_EPS) {
            status[(size_t)c * LN + j] = -1;
            beta[r] = 0; beta[i] = 0; rho[r] = 0; rho[i] = 0;
        } else {
            double const inv = 1. / n2rho;
            beta[r] = (REAL)((zr * rr + zi * ri) * inv);
            beta[i] = (REAL)((zi * rr - zr * ri) * inv);
            rho[r] = (REAL)zr; rho[i] = (REAL)zi;
        }
    }
}

/* alfa = -rho/z, c67 = z*(var*eta/rho); breakdown -> status -2 -- tfQMRdec34, tfqmrgpu_linalg.hxx:116-151 */
void FN(tfqo_dec34)(int LN, uint32_t nCols, int8_t *status, REAL *c67, REAL *alfa, REAL const *rho,
                    REAL const *eta, double const *z, double const *var)
{
    for (uint32_t c = 0; c < nCols; ++c) for (int j = 0; j < LN; ++j) {
        size_t const r = ((size_t)c * 2) * LN + j, i = r + LN;
        double const rr = rho[r], ri = rho[i], zr = z[r], zi = z[i];
        double const n2rho = rr * rr + ri * ri, n2z = zr * zr + zi * zi;
        if (n2z < TFQO_EPS || n2rho < TFQO_EPS) {
            status[(size_t)c * LN + j] = -2;
            alfa[r] = 0; alfa[i] = 0; c67[r] = 0; c67[i] = 0;
        } else {
            double const er = eta[r], ei = eta[i];
            double const minv = -1. / n2z;
            alfa[r] = (REAL)((rr * zr + ri * zi) * minv);
            alfa[i] = (REAL)((ri * zr - rr * zi) * minv);
            double const f = var[(size_t)c * LN + j] / n2rho;
            double const tr = (er * rr + ei * ri) * f, ti = (ei * rr - er * ri) * f;
            c67[r] = (REAL)(zr * tr - zi * ti);
            c67[i] = (REAL)(zi * tr + zr * ti);
        }
    }
}

/* var = d/tau, c = 1/(1+var), tau = d*c, eta = -c*alfa (0 after a breakdown), optional c67 = var*c;
 * |tau| <= eps -> status -3 -- tfQMRdecT, tfqmrgpu_linalg.hxx:195-229 */
void FN(tfqo_decT)(int LN, uint32_t nCols, int8_t *status, REAL *c67 /* may be NULL */, REAL *eta,
                   double *var, double *tau, REAL const *alfa, double const *d55)
{
    for (uint32_t c = 0; c < nCols; ++c) for (int j = 0; j < LN; ++j) {
        size_t const r = ((size_t)c * 2) * LN + j, i = r + LN, s = (size_t)c * LN + j;
        double cosi = 0; REAL r67 = 1;
        double const t = tau[s];
        if (fabs(t) > TFQO_EPS) {
            double const v = d55[s] / t;
            cosi = 1. / (1. + v);
            var[s] = v; tau[s] = d55[s] * cosi;
            r67 = (REAL)(v * cosi);
        } else {
            status[s] = -3; var[s] = 0; tau[s] = 0;
        }
        if (status[s] < 0) { eta[r] = 0; eta[i] = 0; }
        else { eta[r] = (REAL)(-cosi * alfa[r]); eta[i] = (REAL)(-cosi * alfa[i]); }
        if (c67) { c67[r] = r67; c67[i] = 0; }
    }
}

/* test hook (not in the reference): copy of the work vectors v1, v4 ... v9 as they stand at the end of iteration
 * `iteration` (after :233, before the stopping test and its probe), [7][nnzbX*2*LM*LN]; NULL switches it off */
static REAL *FN(dump_buf);
static int FN(dump_iteration);
void FN(tfqo_set_dump)(int iteration, REAL *buf) { FN(dump_iteration) = iteration; FN(dump_buf) = buf; }

/* The tfQMR driver: restates tfqmrgpu::solve (tfqmrgpu_core.hxx:114-325) operation by operation.
 * v3 is supplied by the caller (the reference draws it at setBuffer time).
 * reduce (may be NULL): max-reduces doubles over ranks for the sharded mode (not in the reference).
 * returns 0 converged / 9 out of iterations / 6 all right-hand sides broke down. */
int FN(tfqo_solve)(int LM, int LN, uint32_t nnzbX, uint32_t nnzbB, uint32_t nCols,
    uint32_t const *starts, uint32_t const *pairs, uint32_t const *subset, uint16_t const *colindx,
    REAL const *A, REAL const *B, float const *v3, REAL *X /* out */,
    double tolerance, int MaxIterations,
    int32_t *iterations_needed, double *residuum_reached, double *flops_performed,
    double *bound_history /* [MaxIterations] or NULL */, int32_t *n_history,
    tfqo_reduce_t reduce, void *reduce_ctx)
{
    size_t const E = (size_t)2 * LM * LN, nS = (size_t)nnzbX * E;
    size_t const nR = (size_t)nCols * LN;
    REAL *v1 = X, *v4 = calloc(nS, sizeof(REAL)), *v5 = calloc(nS, sizeof(REAL)), *v6 = calloc(nS, sizeof(REAL)),
         *v7 = calloc(nS, sizeof(REAL)), *v8 = calloc(nS, sizeof(REAL)), *v9 = calloc(nS, sizeof(REAL));
    REAL *rho = calloc(2 * nR, sizeof(REAL)), *alfa = calloc(2 * nR, sizeof(REAL)), *beta = calloc(2 * nR, sizeof(REAL)),
         *c67 = calloc(2 * nR, sizeof(REAL)), *eta = calloc(2 * nR, sizeof(REAL));
    double *zvv = calloc(2 * nR, sizeof(double)), *dvv = calloc(nR, sizeof(double)), *tau = calloc(nR, sizeof(double)),
           *var = calloc(nR, sizeof(double)), *invBn2 = calloc(nR, sizeof(double));
    int8_t *status = calloc(nR, 1);
    for (size_t e = 0; e < nS; ++e) v1[e] = 0;                       /* the initial guess is always zero (:125) */
    for (uint32_t c = 0; c < nCols; ++c) for (int j = 0; j < LN; ++j) rho[((size_t)c * 2) * LN + j] = 1; /* :122 */

    double const blk = (double)LM * LN;
    double const fMult = (double)starts[nnzbX] * 8. * LM * blk, fDot = nnzbX * 8. * blk, fNrm = nnzbX * 4. * blk, fAxp = nnzbX * 8. * blk;
    double nFlop = 0;
    double const tol2 = tolerance * tolerance;
    double target_bound2 = tol2 * 100 * 100, residual2_reached = 1e300;

    FN(tfqo_add_rhs)(LM, LN, nnzbB, subset, v5, B, (REAL)1);         /* v5 := B (:153) */
    FN(tfqo_nrm2)(LM, LN, nnzbX, nCols, colindx, dvv, v5); nFlop += fNrm;
    for (size_t r = 0; r < nR; ++r) { tau[r] = dvv[r]; invBn2[r] = 1. / dvv[r]; } /* :155-166 */

    int ret = 9, iteration = 0, nhist = 0;
    *iterations_needed = MaxIterations;
    while (iteration < MaxIterations) {
        ++iteration;
        FN(tfqo_dotp)(LM, LN, nnzbX, nCols, colindx, zvv, v5, v3); nFlop += fDot;          /* :189 */
        FN(tfqo_dec35)(LN, nCols, status, rho, beta, zvv);                                  /* :192 */
        FN(tfqo_xpay)(LM, LN, nnzbX, colindx, v6, beta, v5); nFlop += fAxp;                 /* :194 */
        FN(tfqo_xpay)(LM, LN, nnzbX, colindx, v4, beta, v8); nFlop += fAxp;                 /* :196 */
        FN(tfqo_spmm)(LM, LN, nnzbX, starts, pairs, A, v6, v9); nFlop += fMult;             /* :198 */
        FN(tfqo_xpay)(LM, LN, nnzbX, colindx, v4, beta, v9); nFlop += fAxp;                 /* :200 */
        FN(tfqo_dotp)(LM, LN, nnzbX, nCols, colindx, zvv, v4, v3); nFlop += fDot;           /* :202 */
        FN(tfqo_dec34)(LN, nCols, status, c67, alfa, rho, eta, zvv, var);                   /* :205 */
        FN(tfqo_xpay)(LM, LN, nnzbX, colindx, v7, c67, v6); nFlop += fAxp;                  /* :207 */
        FN(tfqo_axpy)(LM, LN, nnzbX, colindx, v5, v9, alfa); nFlop += fAxp;                 /* :209 */
        FN(tfqo_nrm2)(LM, LN, nnzbX, nCols, colindx, dvv, v5); nFlop += fNrm;               /* :211 */
        FN(tfqo_decT)(LN, nCols, status, c67, eta, var, tau, alfa, dvv);                    /* :214 */
        FN(tfqo_axpy)(LM, LN, nnzbX, colindx, v1, v7, eta); nFlop += fAxp;                  /* :216 */
        FN(tfqo_axpy)(LM, LN, nnzbX, colindx, v6, v4, alfa); nFlop += fAxp;                 /* :218 */
        FN(tfqo_xpay)(LM, LN, nnzbX, colindx, v7, c67, v6); nFlop += fAxp;                  /* :220 */
        FN(tfqo_spmm)(LM, LN, nnzbX, starts, pairs, A, v6, v8); nFlop += fMult;             /* :224 */
        FN(tfqo_axpy)(LM, LN, nnzbX, colindx, v5, v8, alfa); nFlop += fAxp;                 /* :226 */
        FN(tfqo_nrm2)(LM, LN, nnzbX, nCols, colindx, dvv, v5); nFlop += fNrm;               /* :228 */
        FN(tfqo_decT)(LN, nCols, status, NULL, eta, var, tau, alfa, dvv);                   /* :231 */
        FN(tfqo_axpy)(LM, LN, nnzbX, colindx, v1, v7, eta); nFlop += fAxp;                  /* :233 */

        if (FN(dump_buf) && iteration == FN(dump_iteration)) {
            REAL const *const vs[7] = {v1, v4, v5, v6, v7, v8, v9};
            for (int v = 0; v < 7; ++v) memcpy(FN(dump_buf) + (size_t)v * nS, vs[v], nS * sizeof(REAL));
        }

        /* stopping test on the residual bound (:239-260) */
        double red[2] = {0, 0}; /* {max tau/|b|^2, some RHS not broken down} */
        for (size_t r = 0; r < nR; ++r) {
            double const b2 = tau[r] * invBn2[r];
            if (b2 > red[0]) red[0] = b2;                    /* std::max semantics: NaN never wins */
            if (!(status[r] == -1 || status[r] == -2)) red[1] = 1;
        }
        if (reduce) reduce(reduce_ctx, red, 2);
        double const max_bound2 = red[0] * (2 * iteration + 1);
        if (bound_history) bound_history[nhist] = max_bound2;
        ++nhist;
        int probe = (max_bound2 <= target_bound2 || iteration >= MaxIterations);
        if (red[1] == 0) { iteration += MaxIterations; ret = 6; probe = 0; }

        if (probe) {                                                                         /* :263-304 */
            FN(tfqo_spmm)(LM, LN, nnzbX, starts, pairs, A, v1, v9); nFlop += fMult;
            FN(tfqo_add_rhs)(LM, LN, nnzbB, subset, v9, B, (REAL)-1);
            FN(tfqo_nrm2)(LM, LN, nnzbX, nCols, colindx, dvv, v9); nFlop += fNrm;
            double pr[2] = {0, 0}; /* {max res2, some live RHS above tol2} */
            for (size_t r = 0; r < nR; ++r) {
                double const res2 = dvv[r] * invBn2[r];
                if (res2 > pr[0]) pr[0] = res2;
                if (res2 > tol2) { if (0 == status[r]) pr[1] = 1; }
                else if (res2 <= 0) status[r] = 1;
            }
            if (reduce) reduce(reduce_ctx, pr, 2);
            double const max_residual2 = (pr[0] > 1.4e-76) ? pr[0] : 1.4e-76;
            residual2_reached = max_residual2;
            target_bound2 = (max_bound2 / max_residual2) * tol2;
            if (pr[1] == 0) { *iterations_needed = iteration; iteration += 2 * MaxIterations; ret = 0; }
        }
    }
    *residuum_reached = sqrt(residual2_reached);
    *flops_performed = nFlop;
    if (n_history) *n_history = nhist;
    free(v4); free(v5); free(v6); free(v7); free(v8); free(v9);
    free(rho); free(alfa); free(beta); free(c67); free(eta);
    free(zvv); free(dvv); free(tau); free(var); free(invBn2); free(status);
    return ret;
}

/* block layout conversion user <-> native: restates set_or_getMatrix + transpose_blocks_kernel
 * (tfqmrgpu.cu:489-517,552-562; tfqmrgpu_linalg.hxx:306-360) for one operator.
 * is_A: the stored block is the transpose of op(M).  direction 0: user -> native, 1: native -> user.
 * Rectangular transposed blocks are taken as [nC][nR] arrays (square blocks: same as the reference). */
int FN(tfqo_convert)(int direction, int is_A, uint32_t nnzb, int LM, int LNorLM, int layout, char trans, REAL *native, REAL *user)
{
    int conj = 0, tr = 0;
    switch (trans | 32) {
        case 'h': case 'c': conj = 1; tr = 1; break;
        case '*': conj = 1; break;
        case 't': tr = 1; break;
        case 'n': break;
        default: return 17;
    }
    if (is_A) tr = !tr;
    int const nR = LM, nC = LNorLM;
    size_t const E = (size_t)2 * nR * nC;
    for (uint32_t b = 0; b < nnzb; ++b) for (int c = 0; c < 2; ++c) for (int r = 0; r < nR; ++r) for (int s = 0; s < nC; ++s) {
        int const cols = tr ? nR : nC, i = tr ? s : r, j = tr ? r : s;
        size_t uo;
        switch (layout) {
            case 0x0f: uo = (size_t)c * nR * nC + (size_t)i * cols + j; break;
            case 0x33: uo = (size_t)i * 2 * cols + (size_t)c * cols + j; break;
            case 0x55: uo = ((size_t)i * cols + j) * 2 + c; break;
            default: return 15;
        }
        size_t const no = (size_t)c * nR * nC + (size_t)r * nC + s;
        REAL const sign = (conj && c) ? (REAL)-1 : (REAL)1;
        if (0 == direction) native[b * E + no] = sign * user[b * E + uo];
        else                user[b * E + uo] = sign * native[b * E + no];
    }
    return 0;
}

#undef FN
#undef CAT
#undef CAT_
