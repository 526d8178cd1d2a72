// Calibration of the matrix pipe of gfx950 for the two instructions the multiply kernels use (and the 32x32x2 f32 form):
// cycles per instruction measured INSIDE the kernel (s_memtime = shader clock, s_memrealtime = 100 MHz), one wave per
// SIMD (256-thread work groups, one per CU) or more, NACC independent accumulators, 64 back-to-back MFMAs per loop trip
// so that the loop branch is < 1 % of the stream, operands in registers, random non-trivial data.
// build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_rate.hip -o /tmp/mfma_rate     (round-1 probe: scripts/mfma_peak.hip,
// 4 MFMAs per loop trip -- the loop overhead made it read low)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
using d4 = __attribute__((ext_vector_type(4))) double;
using f4 = __attribute__((ext_vector_type(4))) float;
using f16v = __attribute__((ext_vector_type(16))) float;

struct Stamp { unsigned long long c0, c1, r0, r1; };

template <int KIND, int NACC> __global__ __launch_bounds__(256) void k_rate(Stamp* st, double* out, int trips, float seed) {
    constexpr int PER_TRIP = 64;
    double const a = 0.37 + threadIdx.x * 1e-3 + seed, b = 1.0 / (1.0 + threadIdx.x * 1e-2);
    float const af = float(a), bf = float(b);
    d4 cd[NACC]; f4 cf[NACC]; f16v cw[NACC];
    for (int i = 0; i < NACC; ++i) { cd[i] = d4{0, 0, 0, 0}; cf[i] = f4{0, 0, 0, 0}; for (int e = 0; e < 16; ++e) cw[i][e] = 0; }
    __syncthreads();
    unsigned long long const c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int u = 0; u < PER_TRIP; ++u) {
            if constexpr (KIND == 0) cd[u % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cd[u % NACC], 0, 0, 0);
            if constexpr (KIND == 1) cf[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, cf[u % NACC], 0, 0, 0);
            if constexpr (KIND == 2) cw[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, cw[u % NACC], 0, 0, 0);
            if constexpr (KIND == 3) cd[u % NACC][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, cd[u % NACC][0], 0, 0, 0);
        }
    }
    unsigned long long const c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) { s += cd[i][0] + cd[i][1] + cd[i][2] + cd[i][3] + cf[i][0] + cf[i][1] + cf[i][2] + cf[i][3]; for (int e = 0; e < 16; ++e) s += cw[i][e]; }
    out[blockIdx.x * size_t(blockDim.x) + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1};
}

template <int KIND, int NACC> void run(char const* name, double flopPerInstr, int wgs, Stamp* dst, double* out) {
    int const trips = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) k_rate<KIND, NACC><<<wgs, 256>>>(dst, out, trips, 0.01f * w);   // ~ a second of load before the timed launch (DVFS settles)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<KIND, NACC><<<wgs, 256>>>(dst, out, trips, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(size_t(wgs) * 4);
    hipMemcpy(h.data(), dst, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (auto const& s : h) {
        cyc.push_back(double(s.c1 - s.c0) / (double(trips) * 64));
        ghz.push_back(double(s.c1 - s.c0) / (double(s.r1 - s.r0) * 10.0));   // 100 MHz ticks -> ns
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    double const instr = double(wgs) * 4 * trips * 64;
    int const perSimd = std::max(1, wgs / 256);
    printf("%-22s %2d acc, %4d WGs (%d wave%s per SIMD): %.3f ms  %7.1f TFLOP/s | in-kernel: median %.1f clk per instruction and wave"
           " -> %.1f clk per SIMD, shader clock %.2f GHz\n", name, NACC, wgs, perSimd, perSimd > 1 ? "s" : "", ms,
           instr * flopPerInstr / ms * 1e-9, cyc[cyc.size() / 2], cyc[cyc.size() / 2] / perSimd, ghz[ghz.size() / 2]);
}

int main() {
    Stamp* st; double* out;
    hipMalloc(&st, 8192 * 4 * sizeof(Stamp)); hipMalloc(&out, size_t(8192) * 256 * 8);
    for (int wgs : {256, 512, 1024, 2048}) {
        run<0, 4>("v_mfma_f64_16x16x4", 2048, wgs, st, out);
        run<0, 8>("v_mfma_f64_16x16x4", 2048, wgs, st, out);
        run<1, 4>("v_mfma_f32_16x16x4", 2048, wgs, st, out);
        run<1, 8>("v_mfma_f32_16x16x4", 2048, wgs, st, out);
        run<2, 4>("v_mfma_f32_32x32x2", 4096, wgs, st, out);
        run<3, 8>("v_mfma_f64_4x4x4_4b", 512, wgs, st, out);
    }
    return 0;
}
