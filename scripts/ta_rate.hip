// micro-benchmark: cache-resident operand loads at 8 B/lane against 16 B/lane, alone and interleaved with
// f64 MFMAs at the ratio of the 16x16 multiply (1 load : 1 MFMA at 8 B, 1 : 2 at 16 B).
// build: hipcc --offload-arch=gfx950 -O3 scripts/ta_rate.hip -o /tmp/ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using d4 = __attribute__((ext_vector_type(4))) double;

template <int W, bool MMA> __global__ __launch_bounds__(256) void k_load(double const* __restrict__ src, double* out, int iters, int blocks) {
    // every wave walks over `blocks` 4-KiB blocks that stay cache resident
    int const lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d4 c[4]; for (int i = 0; i < 4; ++i) c[i] = d4{0, 0, 0, 0};
    double s = 0;
    unsigned b = (blockIdx.x * 4 + wave) * 7u;
    for (int it = 0; it < iters; ++it) {
        double const* p = src + size_t(b % blocks) * 512; b += 13;
        if constexpr (W == 8) {
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = p[j * 64 + lane];
#pragma unroll
            for (int j = 0; j < 8; ++j) { if constexpr (MMA) c[j & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[j], v[j ^ 1], c[j & 3], 0, 0, 0); else s += v[j]; }
        } else {
            double2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = reinterpret_cast<double2 const*>(p)[j * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (MMA) { c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[j].x, v[j ^ 1].x, c[j], 0, 0, 0);
                                     c[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(v[j].y, v[j ^ 1].y, c[j], 0, 0, 0); }
                else s += v[j].x + v[j].y;
            }
        }
    }
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    double *src, *out; hipMalloc(&src, 64 << 20); hipMalloc(&out, 64 << 20); hipMemset(src, 0, 64 << 20);
    int const iters = 2000, wg = 2048;
    for (int blocks : {4, 256, 4096}) {           // 16 KiB (L1), 1 MiB (L2), 16 MiB (L2 of all XCDs / MALL)
        double bytes = double(wg) * 4 * iters * 4096.0, flops = double(wg) * 4 * iters * 8 * 2048.0;
        float ms;
        ms = timeit([&] { k_load<8, false><<<wg, 256>>>(src, out, iters, blocks); }, 3);
        printf("%5d blocks  8 B/lane loads only : %.3f ms  %.1f TB/s  %.1f B/clk/CU\n", blocks, ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 256 / 2.4 * 1e3 * 1e-3);
        ms = timeit([&] { k_load<16, false><<<wg, 256>>>(src, out, iters, blocks); }, 3);
        printf("%5d blocks 16 B/lane loads only : %.3f ms  %.1f TB/s  %.1f B/clk/CU\n", blocks, ms, bytes / ms * 1e-9, bytes / ms * 1e-9 / 256 / 2.4 * 1e3 * 1e-3);
        ms = timeit([&] { k_load<8, true><<<wg, 256>>>(src, out, iters, blocks); }, 3);
        printf("%5d blocks  8 B/lane + MFMA 1:1  : %.3f ms  %.1f TB/s  %.1f TFLOP/s\n", blocks, ms, bytes / ms * 1e-9, flops / ms * 1e-9);
        ms = timeit([&] { k_load<16, true><<<wg, 256>>>(src, out, iters, blocks); }, 3);
        printf("%5d blocks 16 B/lane + MFMA 1:2  : %.3f ms  %.1f TB/s  %.1f TFLOP/s\n", blocks, ms, bytes / ms * 1e-9, flops / ms * 1e-9);
    }
    return 0;
}
