// which XCD does work group b of a 1-D grid land on?  (reads HW_REG_XCC_ID; gfx942/gfx950)
// build: hipcc --offload-arch=gfx950 -O3 scripts/xcc_map.hip -o /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out, int spin) {
    unsigned const id = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // XCC_ID[3:0]
    if (threadIdx.x == 0) out[blockIdx.x] = id;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
    int const n = 4096;
    unsigned* d; hipMalloc(&d, n * 4);
    for (int threads : {64, 256, 512}) {
        k<<<n, threads>>>(d, 200); hipDeviceSynchronize();
        std::vector<unsigned> h(n); hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        int agree = 0; for (int b = 0; b < n; ++b) agree += (h[b] == unsigned(b % 8));
        printf("%d threads: first 32 work groups ->", threads);
        for (int b = 0; b < 32; ++b) printf(" %u", h[b]);
        printf("\n  work groups with XCC_ID == blockIdx %% 8: %d of %d\n", agree, n);
    }
    return 0;
}
