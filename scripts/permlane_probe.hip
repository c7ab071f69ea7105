// probe: what v_permlane16_swap_b32 does to its two operands (gfx950)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/_prof/permlane_probe scripts/permlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    unsigned a = 100 + threadIdx.x, b = 200 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("a = 100 + lane, b = 200 + lane; (x, y) = permlane16_swap(a, b)\n");
    for (int row = 0; row < 4; row++) printf("row %d (lanes %2d..): x = %u.., y = %u..\n", row, 16 * row, h[16 * row], h[64 + 16 * row]);
    return 0;
}
