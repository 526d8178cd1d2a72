// r04 probe: what a device-wide barrier costs inside ONE cooperative launch (cooperative_groups grid.sync()) against the boundary between two dependent
// launches -- the question behind a persistent "iteration slot" kernel for small systems.
// build: hipcc -O3 --offload-arch=gfx950 scripts/grid_sync_probe.hip -o scripts/bin/grid_sync_probe ; run: scripts/bin/grid_sync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_syncs(double* buf, int n, int rounds) {
    cg::grid_group g = cg::this_grid();
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = buf[i % n];
    for (int r = 0; r < rounds; ++r) {
        buf[i % n] = v + 1.0;              // a little memory traffic per phase, as a phase of a slot would have
        g.sync();
        v = buf[(i + 256) % n];
    }
    buf[i % n] = v;
}
__global__ __launch_bounds__(256) void k_phase(double* buf, int n) {
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    buf[i % n] = buf[(i + 256) % n] + 1.0;
}

int main() {
    int const n = 1 << 20;
    double* buf; CK(hipMalloc(&buf, n * sizeof(double))); CK(hipMemset(buf, 0, n * sizeof(double)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wgs : {64, 256, 512, 768}) {
        int rounds = 200;
        int nn = n;
        void* args[] = { &buf, &nn, &rounds };
        hipError_t le = hipLaunchCooperativeKernel((void const*)k_syncs, dim3(wgs), dim3(256), args, 0, s);
        if (le != hipSuccess) { std::printf("%d work groups: cooperative launch refused (%s)\n", wgs, hipGetErrorString(le)); (void)hipGetLastError(); continue; }
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        CK(hipLaunchCooperativeKernel((void const*)k_syncs, dim3(wgs), dim3(256), args, 0, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < rounds; ++r) k_phase<<<dim3(wgs), dim3(256), 0, s>>>(buf, n);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
        std::printf("%4d work groups: grid.sync() %.2f us per phase (one cooperative launch of %d phases) | %.2f us per phase as %d dependent launches\n",
                    wgs, ms * 1e3 / rounds, rounds, ms2 * 1e3 / rounds, rounds);
    }
    return 0;
}
