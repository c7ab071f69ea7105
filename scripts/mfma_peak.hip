// probe: sustained v_mfma_f32_16x16x4_f32 rate of the whole chip (no memory traffic), for the roofline's peak
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/_prof/mfma_peak scripts/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; c++) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
void run(int wg_per_cu, int iters) {
    float *out; hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
    int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * CHAINS * 2048.0;
    printf("chains %d, %d workgroups/CU (%d waves/SIMD): %.3f ms, %.1f TFLOP/s\n", CHAINS, wg_per_cu, wg_per_cu, ms, flop / ms / 1e9);
    hipFree(out);
}
int main() {
    // short bursts (a few ms) and sustained runs (>= 50 ms: the clock settles under the matrix-core load)
    run<8>(1, 20000); run<8>(2, 20000); run<4>(3, 20000);
    run<8>(1, 400000); run<8>(2, 200000); run<4>(3, 300000); run<2>(3, 600000);
    return 0;
}
