// probe: cycles per K-step of the split tower's inner loop (6 x v_mfma_f32_16x16x32_f16 + 4 x ds_read_b128 per wave), one
// or two workgroups of 4 waves per CU, by accumulator pattern and LDS load placement.  What limits k_trunk16s's convs?
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/_prof/mfma16_pattern scripts/mfma16_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)

// MODE 0: six independent accumulators, no LDS     1: the tower's pattern (m0 m1 c0 c1 c0 c1), no LDS
// MODE 2: pattern + 4 ds_read_b128 one step ahead   3: pattern + loads two steps ahead
// MODE 4: loads only (no MFMA)                      5: pattern m0 c0 m1 c1 c0 c1 interleaved + loads one step ahead
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[20480];
    for (int i = threadIdx.x; i < 20480; i += 256) lds[i] = (_Float16)(0.001f * (i & 255));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const _Float16 *p = lds + wave * 4096 + (lane & 15) * 16 + ((lane >> 4) & 1) * 8 + (lane >> 5) * 16;
    f16x8 wh = *(const f16x8 *)(p), wl = *(const f16x8 *)(p + 512);
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
    f16x8 h0[3], h1[3], l0[3], l1[3];
    for (int j = 0; j < 3; j++) { h0[j] = wh; h1[j] = wl; l0[j] = wh; l1[j] = wl; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int cur = u % 3, nxt = (MODE == 3) ? (u + 2) % 3 : (u + 1) % 3;
            if (MODE == 6 || MODE == 7) {
                constexpr int D = MODE == 7 ? 2 : 1;
                const int nx = (u + D) % 3;
                const _Float16 *q = p + (u & 3) * 256;
                a0 = MF(wh, h0[cur], a0);
                __builtin_amdgcn_sched_barrier(0);
                h0[nx] = *(const f16x8 *)(q);
                __builtin_amdgcn_sched_barrier(0);
                a1 = MF(wh, h1[cur], a1);
                __builtin_amdgcn_sched_barrier(0);
                l0[nx] = *(const f16x8 *)(q + 208);
                __builtin_amdgcn_sched_barrier(0);
                a2 = MF(wh, l0[cur], a2);
                __builtin_amdgcn_sched_barrier(0);
                h1[nx] = *(const f16x8 *)(q + 1024);
                __builtin_amdgcn_sched_barrier(0);
                a3 = MF(wh, l1[cur], a3);
                __builtin_amdgcn_sched_barrier(0);
                l1[nx] = *(const f16x8 *)(q + 1024 + 208);
                __builtin_amdgcn_sched_barrier(0);
                a2 = MF(wl, h0[cur], a2); a3 = MF(wl, h1[cur], a3);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            if (MODE >= 2) {
                const _Float16 *q = p + (u & 3) * 256;
                h0[nxt] = *(const f16x8 *)(q); l0[nxt] = *(const f16x8 *)(q + 208);
                h1[nxt] = *(const f16x8 *)(q + 1024); l1[nxt] = *(const f16x8 *)(q + 1024 + 208);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 0) {
                a0 = MF(wh, h0[cur], a0); a1 = MF(wh, h1[cur], a1); a2 = MF(wh, l0[cur], a2);
                a3 = MF(wh, l1[cur], a3); a4 = MF(wl, h0[cur], a4); a5 = MF(wl, h1[cur], a5);
            } else if (MODE == 5) {
                a0 = MF(wh, h0[cur], a0); a2 = MF(wh, l0[cur], a2); a1 = MF(wh, h1[cur], a1);
                a3 = MF(wh, l1[cur], a3); a2 = MF(wl, h0[cur], a2); a3 = MF(wl, h1[cur], a3);
            } else if (MODE != 4) {
                a0 = MF(wh, h0[cur], a0); a1 = MF(wh, h1[cur], a1); a2 = MF(wh, l0[cur], a2);
                a3 = MF(wh, l1[cur], a3); a2 = MF(wl, h0[cur], a2); a3 = MF(wl, h1[cur], a3);
            } else {
                a0[0] += (float)h0[cur][0] + (float)l0[cur][0] + (float)h1[cur][0] + (float)l1[cur][0];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = a0 + a1 + a2 + a3 + a4 + a5;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char *name, int wg_per_cu) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 2 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %d wg/CU: %6.1f cycles per K-step (6 MFMA = 96 at peak); %.3f ms, %.0f TFLOP/s\n", name, wg_per_cu, (double)c / (iters * 6.0),
           ms, MODE == 4 ? 0.0 : 256.0 * wg_per_cu * 4 * iters * 36.0 * 16384.0 / ms / 1e9);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w = 1; w <= 2; w++) {
        run<0>("six independent accumulators, operands in registers", w);
        run<1>("tower pattern m0 m1 c0 c1 c0 c1, operands in registers", w);
        run<5>("pattern m0 c0 m1 c1 c0 c1 + 4 ds_read_b128, 1 step ahead", w);
        run<2>("tower pattern + 4 ds_read_b128, 1 step ahead", w);
        run<3>("tower pattern + 4 ds_read_b128, 2 steps ahead", w);
        run<4>("4 ds_read_b128 per step only", w);
        run<6>("tower pattern, one ds_read_b128 after each of MFMA 1-4, 1 ahead", w);
        run<7>("tower pattern, one ds_read_b128 after each of MFMA 1-4, 2 ahead", w);
    }
    return 0;
}
