// Lane layout of v_mfma_f64_4x4x4_4b_f64 (four 4 x 4 x 4 products per instruction), found by one-hot operands:
// for every pair (la, lb) of operand lanes the lanes of D that receive A[la] * B[lb].  hipcc --offload-arch=gfx950 -O2 mfma4_probe.hip -o bin/mfma4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double* out) {
    int const lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double const a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
            double const d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            out[(size_t(la) * 64 + lb) * 64 + lane] = d;
        }
}
int main() {
    double* d; size_t n = 64 * 64 * 64;
    if (hipMalloc(&d, n * 8) != hipSuccess) return 1;
    k<<<1, 64>>>(d);
    std::vector<double> h(n);
    if (hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    // per A lane: the B lanes it meets and the D lanes the products land in
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            for (int ld = 0; ld < 64; ++ld)
                if (h[(size_t(la) * 64 + lb) * 64 + ld] != 0.0) printf("  B%d->D%d", lb, ld);
        printf("\n");
    }
    return 0;
}
