// Shader clock under load, measured inside the kernels: clock64() (s_memtime, shader clock) against wall_clock64()
// (s_memrealtime, constant 100 MHz).  Is the f64 MFMA rate of ~49 TFLOP/s (spec 78.6) a clock that drops under matrix load?
// build: hipcc --offload-arch=gfx950 -O3 scripts/clock_in_kernel.hip -o /tmp/clock_in_kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;
struct Rec { long long c, w; };

template <int NACC, bool F64> __global__ __launch_bounds__(256) void k_mfma(double* out, Rec* rec, int iters) {
    long long const c0 = clock64(), w0 = wall_clock64();
    double s = 0;
    if constexpr (F64) {
        d4 c[NACC];
        for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f4 c[NACC];
        for (int i = 0; i < NACC; ++i) c[i] = f4{0, 0, 0, 0};
        float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    long long const c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{c1 - c0, w1 - w0};
}
// pure VALU fp64 FMA loop (no matrix pipe)
__global__ __launch_bounds__(256) void k_valu(double* out, Rec* rec, int iters) {
    long long const c0 = clock64(), w0 = wall_clock64();
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9, c[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = __builtin_fma(a, c[i], b);
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += c[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    long long const c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{c1 - c0, w1 - w0};
}
using d2 = __attribute__((ext_vector_type(2))) double;
__global__ __launch_bounds__(256) void k_copy(d2* __restrict__ y, d2 const* __restrict__ x, size_t n, Rec* rec) {
    long long const c0 = clock64(), w0 = wall_clock64();
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
    long long const c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) rec[blockIdx.x] = Rec{c1 - c0, w1 - w0};
}
static void report(char const* what, Rec* drec, int n, float ms, double flops) {
    std::vector<Rec> r(n); hipMemcpy(r.data(), drec, n * sizeof(Rec), hipMemcpyDeviceToHost);
    std::vector<double> ghz; for (auto& x : r) if (x.w > 0) ghz.push_back(double(x.c) / double(x.w) * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("%-34s %8.3f ms  %7.1f TFLOP/s   clock64/wall_clock64: median %.3f GHz (min %.3f max %.3f), median duration %.1f us\n", what, ms, flops / ms * 1e-9,
           ghz[ghz.size() / 2], ghz.front(), ghz.back(), r[n / 2].w * 0.01);
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    double* out; hipMalloc(&out, size_t(1) << 28);
    Rec* rec; hipMalloc(&rec, sizeof(Rec) << 20);
    int const iters = 20000;
    for (int wg : {256, 1024, 2048}) {
        float ms = timeit([&] { k_mfma<4, true><<<wg, 256>>>(out, rec, iters); }, 3);
        char buf[64]; snprintf(buf, 64, "mfma f64 %4d WGs, 4 acc", wg);
        report(buf, rec, wg, ms, double(wg) * 4 * iters * 4 * 2048.0);
        ms = timeit([&] { k_mfma<4, false><<<wg, 256>>>(out, rec, iters); }, 3);
        snprintf(buf, 64, "mfma f32 %4d WGs, 4 acc", wg);
        report(buf, rec, wg, ms, double(wg) * 4 * iters * 4 * 2048.0);
    }
    {   // one wave per SIMD only, a single accumulator chain: the latency of one MFMA
        float ms = timeit([&] { k_mfma<1, true><<<256, 256>>>(out, rec, iters); }, 3);
        report("mfma f64 256 WGs, 1 acc (latency)", rec, 256, ms, 256.0 * 4 * iters * 2048.0);
        ms = timeit([&] { k_mfma<8, true><<<256, 256>>>(out, rec, iters); }, 3);
        report("mfma f64 256 WGs, 8 acc", rec, 256, ms, 256.0 * 4 * iters * 8 * 2048.0);
        ms = timeit([&] { k_mfma<8, true><<<2048, 256>>>(out, rec, iters / 4); }, 3);
        report("mfma f64 2048 WGs, 8 acc", rec, 2048, ms, 2048.0 * 4 * (iters / 4) * 8 * 2048.0);
    }
    {
        float ms = timeit([&] { k_valu<<<2048, 256>>>(out, rec, iters); }, 3);
        report("valu f64 fma 2048 WGs", rec, 2048, ms, 2048.0 * 256 * iters * 8 * 2.0);
    }
    size_t const n = size_t(1) << 26;
    d2 *x, *y; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMemset(x, 0, n * 16);
    float ms = timeit([&] { k_copy<<<65536, 256>>>(y, x, n, rec); }, 5);
    report("copy nt 2 GiB, 65536 WGs", rec, 65536, ms, 0);
    printf("   copy: %.0f GB/s\n", 2.0 * n * 16 / ms * 1e-6);
    return 0;
}
