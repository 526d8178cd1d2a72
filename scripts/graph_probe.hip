// What a dependent kernel boundary costs inside a hipGraph against plain stream launches (the slot of a small system is 8 dependent launches of ~7 us each):
// the same 8 kernels x 200 slots (a) launched one by one on a stream, (b) captured once into a graph of 8 kernel nodes and launched 200 times.
// hipcc --offload-arch=gfx950 -O2 graph_probe.hip -o bin/graph_probe ; usage: bin/graph_probe [work groups] [busy loop length]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_step(double* v, int n, int spin) {
    int const i = blockIdx.x * blockDim.x + threadIdx.x;
    double x = v[i % n];
    for (int s = 0; s < spin; ++s) x = x * 1.0000001 + 1e-9;
    v[i % n] = x;
}
int main(int argc, char** argv) {
    int const wgs = argc > 1 ? atoi(argv[1]) : 256, spin = argc > 2 ? atoi(argv[2]) : 0, slots = 200, per = 8;
    double* v; CK(hipMalloc(&v, size_t(wgs) * 256 * 8)); CK(hipMemset(v, 0, size_t(wgs) * 256 * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto slot = [&]() { for (int k = 0; k < per; ++k) k_step<<<wgs, 256, 0, s>>>(v, wgs * 256, spin); };
    for (int i = 0; i < 20; ++i) slot();
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < slots; ++i) slot();
    CK(hipEventRecord(e1, s));
    auto t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double const enq = std::chrono::duration<double, std::micro>(t1 - t0).count();
    printf("%4d work groups, spin %d: stream launches  %.2f us per kernel on the GPU, %.2f us per kernel to enqueue\n", wgs, spin, ms * 1e3 / (slots * per), enq / (slots * per));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    slot();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < slots; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    t1 = std::chrono::steady_clock::now();
    CK(hipStreamSynchronize(s));
    CK(hipEventElapsedTime(&ms, e0, e1));
    double const enq2 = std::chrono::duration<double, std::micro>(t1 - t0).count();
    printf("%4d work groups, spin %d: graph of %d nodes   %.2f us per kernel on the GPU, %.2f us per kernel to enqueue\n", wgs, spin, per, ms * 1e3 / (slots * per), enq2 / (slots * per));
    return 0;
}
