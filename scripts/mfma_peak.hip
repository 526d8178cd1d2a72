// micro-benchmarks that calibrate the roofs on the GPU box: f64/f32 16x16x4 MFMA issue rate and a
// 16-B-per-lane streaming copy / triad.  build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;

template <int NACC> __global__ __launch_bounds__(256) void k_mfma_f64(double* out, int iters) {
    d4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ __launch_bounds__(256) void k_mfma_f32(float* out, int iters) {
    f4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_copy(double2* __restrict__ y, double2 const* __restrict__ x, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) y[i] = x[i];
}
__global__ __launch_bounds__(256) void k_triad(double2* __restrict__ y, double2 const* __restrict__ x, double2 const* __restrict__ z, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        double2 a = x[i], b = z[i], c = y[i];
        c.x = a.x + 1.5 * b.x - 0.5 * c.x; c.y = a.y + 1.5 * b.y - 0.5 * c.y;
        y[i] = c;
    }
}
using d2 = __attribute__((ext_vector_type(2))) double;
// the same streams with non-temporal accesses (what the solver's vector kernels use)
__global__ __launch_bounds__(256) void k_copy_nt(d2* __restrict__ y, d2 const* __restrict__ x, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
}
__global__ __launch_bounds__(256) void k_triad_nt(d2* __restrict__ y, d2 const* __restrict__ x, d2 const* __restrict__ z, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        d2 const a = __builtin_nontemporal_load(x + i), b = __builtin_nontemporal_load(z + i), c = __builtin_nontemporal_load(y + i);
        __builtin_nontemporal_store(a + 1.5 * b - 0.5 * c, y + i);
    }
}
// 4 read streams + 3 write streams, the shape of k_x_v6_v7
__global__ __launch_bounds__(256) void k_seven_nt(d2* __restrict__ a, d2* __restrict__ b, d2* __restrict__ c, d2 const* __restrict__ d, size_t n) {
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        d2 const va = __builtin_nontemporal_load(a + i), vb = __builtin_nontemporal_load(b + i), vc = __builtin_nontemporal_load(c + i), vd = __builtin_nontemporal_load(d + i);
        __builtin_nontemporal_store(va + 0.5 * vd, a + i);
        __builtin_nontemporal_store(vb + 0.25 * va, b + i);
        __builtin_nontemporal_store(vc - 0.5 * vb, c + i);
    }
}
template <class F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < reps; ++r) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
int main() {
    double* out; hipMalloc(&out, 1 << 26);
    int const iters = 4000;
    for (int wg : {256, 512, 1024, 2048}) {
        float ms = timeit([&] { k_mfma_f64<4><<<wg, 256>>>(out, iters); }, 5);
        double flops = double(wg) * 4 * iters * 4 * 2048.0;
        printf("mfma f64 16x16x4: %4d WGs x 4 waves, 4 acc: %.3f ms  %.1f TFLOP/s\n", wg, ms, flops / ms * 1e-9);
        ms = timeit([&] { k_mfma_f32<4><<<wg, 256>>>((float*)out, iters); }, 5);
        printf("mfma f32 16x16x4: %4d WGs x 4 waves, 4 acc: %.3f ms  %.1f TFLOP/s\n", wg, ms, flops / ms * 1e-9);
    }
    size_t const n = size_t(1) << 26; // 64 Mi double2 = 1 GiB per array
    double2 *x, *y, *z; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16); hipMalloc(&z, n * 16);
    hipMemset(x, 0, n * 16); hipMemset(y, 0, n * 16); hipMemset(z, 0, n * 16);
    for (int wg : {2048, 4096, 16384, 65536}) {
        float ms = timeit([&] { k_copy<<<wg, 256>>>(y, x, n); }, 5);
        printf("copy  16 B/lane, %6d WGs: %.3f ms  %.0f GB/s\n", wg, ms, 2.0 * n * 16 / ms * 1e-6);
        ms = timeit([&] { k_triad<<<wg, 256>>>(y, x, z, n); }, 5);
        printf("triad 16 B/lane, %6d WGs: %.3f ms  %.0f GB/s\n", wg, ms, 4.0 * n * 16 / ms * 1e-6);
    }
    double2* w; hipMalloc(&w, n * 16); hipMemset(w, 0, n * 16);
    for (int wg : {2048, 16384, 65536, 262144}) {
        float ms = timeit([&] { k_copy_nt<<<wg, 256>>>((d2*)y, (d2 const*)x, n); }, 5);
        printf("copy  16 B/lane non-temporal, %6d WGs: %.3f ms  %.0f GB/s\n", wg, ms, 2.0 * n * 16 / ms * 1e-6);
        ms = timeit([&] { k_triad_nt<<<wg, 256>>>((d2*)y, (d2 const*)x, (d2 const*)z, n); }, 5);
        printf("triad 16 B/lane non-temporal, %6d WGs: %.3f ms  %.0f GB/s\n", wg, ms, 4.0 * n * 16 / ms * 1e-6);
        ms = timeit([&] { k_seven_nt<<<wg, 256>>>((d2*)x, (d2*)y, (d2*)z, (d2 const*)w, n); }, 5);
        printf("4 in + 3 out non-temporal,    %6d WGs: %.3f ms  %.0f GB/s\n", wg, ms, 7.0 * n * 16 / ms * 1e-6);
    }
    return 0;
}
