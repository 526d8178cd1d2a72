/* A C caller of libtfQMRgpu.so written the way an existing user of the reference library is: its own stream typedef,
 * the unchanged prototypes of tfqmrgpu.h, host arrays handed to the one-call driver tfqmrgpu_bsrsv_z
 * (reference: tfQMRgpu/include/tfqmrgpu.h:119-156; calling convention of example/tfqmrgpu_C_example.c:17).
 * The system is the known-answer test of the reference's Julia example (example/tfqmrgpu_Julia_example.jl:41-66,117-120):
 * 1-D finite-difference operator (2,-1) (x) I_4 on mb = 7 block rows, B = e_7 with the phases i^p, blocks 4 x 5;
 * the solution is the straight line k/8 times the phase.  Pure C99, no HIP header needed. */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef size_t cudaStream_t;                /* what the legacy caller already has */
#define TFQMRGPU_STREAM_TYPE cudaStream_t   /* the one new line   */
#include "tfqmrgpu.h"

enum { MB = 7, LDA = 4, LDB = 5 };

int main(void) {
    int32_t rowPtrA[MB + 1] = {0}, rowPtrX[MB + 1] = {0}, rowPtrB[MB + 1] = {0};
    int32_t colIndA[3 * MB], colIndX[MB] = {0}, colIndB[1] = {0};
    double *A = calloc((size_t)3 * MB * LDA * LDA * 2, sizeof(double));     /* [nnzbA][LDA][LDA][Re,Im] */
    double *X = calloc((size_t)MB * LDA * LDB * 2, sizeof(double));
    double B[LDA * LDB * 2] = {0};                                          /* [1][LDA][LDB][Re,Im] */
    int nnzbA = 0;
    for (int ib = 0; ib < MB; ++ib) {
        for (int jb = (ib > 0 ? ib - 1 : 0); jb <= (ib < MB - 1 ? ib + 1 : MB - 1); ++jb) {
            colIndA[nnzbA] = jb;
            for (int i = 0; i < LDA; ++i) A[(((size_t)nnzbA * LDA + i) * LDA + i) * 2] = (ib == jb) ? 2.0 : -1.0;
            ++nnzbA;
        }
        rowPtrA[ib + 1] = nnzbA; rowPtrX[ib + 1] = ib + 1;
    }
    rowPtrB[MB] = 1;                                                        /* the only block of B sits in the last row */
    for (int j = 0; j < LDB; ++j) {                                         /* B[j % LDA][j] = i^(j / LDA) */
        int const p = j / LDA;
        B[((j % LDA) * LDB + j) * 2 + 0] = (p % 4 == 0) - (p % 4 == 2);
        B[((j % LDA) * LDB + j) * 2 + 1] = (p % 4 == 1) - (p % 4 == 3);
    }
    int32_t iterations = 210; float residual = 1.2e-8f;
    tfqmrgpuStatus_t const st = tfqmrgpu_bsrsv_z(MB, LDA, LDB,
        rowPtrA, nnzbA, colIndA, A, 'n', rowPtrX, MB, colIndX, X, 'n', rowPtrB, 1, colIndB, B, 'n',
        &iterations, &residual, 0, 0);
    if (st != TFQMRGPU_STATUS_SUCCESS) { tfqmrgpuPrintError(st); printf("c_caller: status %d\n", (int)st); return 2; }
    double maxdev = 0;
    for (int ib = 0; ib < MB; ++ib)
        for (int j = 0; j < LDB; ++j) {
            int const p = j / LDA;
            double const line = (ib + 1) / 8.0;
            double const er = line * ((p % 4 == 0) - (p % 4 == 2)), ei = line * ((p % 4 == 1) - (p % 4 == 3));
            double const *x = X + (((size_t)ib * LDA + j % LDA) * LDB + j) * 2;
            maxdev = fmax(maxdev, fmax(fabs(x[0] - er), fabs(x[1] - ei)));
        }
    printf("c_caller: %d iterations, residual %.3e, max deviation from k/8 %.3e\n", (int)iterations, residual, maxdev);
    if (!(maxdev < 1e-9) || iterations < 1 || iterations > 210) return 1;
    puts("c_caller: OK");
    free(A); free(X);
    return 0;
}
