// Vector-ALU issue rate of one gfx950 SIMD for the instruction classes the legal-move search is made of (the k_rollout
// roofline: it is bound by vector-instruction issue, not by HBM).  W waves per SIMD each run a stream of independent
// wave64 instructions; prints cycles per wave-instruction and SIMD.  (generated body: one asm block of 8 independent
// instructions, repeated 8 times per loop iteration)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_prof/valu_peak scripts/valu_peak.hip && scripts/_prof/valu_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5, s6 = s0 + 6, s7 = s0 + 7;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 1) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %8\n v_fma_f32 %1, %1, %1, %8\n v_fma_f32 %2, %2, %2, %8\n v_fma_f32 %3, %3, %3, %8\n v_fma_f32 %4, %4, %4, %8\n v_fma_f32 %5, %5, %5, %8\n v_fma_f32 %6, %6, %6, %8\n v_fma_f32 %7, %7, %7, %8\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(f0));) }
        else if (KIND == 2) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 3) { REP8(asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 4) { REP8(asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 5) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 6) { REP8(asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 7) { REP8(asm volatile("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3\n v_lshrrev_b32 %4, 1, %4\n v_lshrrev_b32 %5, 1, %5\n v_lshrrev_b32 %6, 1, %6\n v_lshrrev_b32 %7, 1, %7\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 8) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 9) { REP8(asm volatile("v_bfe_u32 %0, %0, 1, 31\n v_bfe_u32 %1, %1, 1, 31\n v_bfe_u32 %2, %2, 1, 31\n v_bfe_u32 %3, %3, 1, 31\n v_bfe_u32 %4, %4, 1, 31\n v_bfe_u32 %5, %5, 1, 31\n v_bfe_u32 %6, %6, 1, 31\n v_bfe_u32 %7, %7, 1, 31\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 10) { REP8(asm volatile("v_and_or_b32 %0, %0, %8, %1\n v_and_or_b32 %1, %1, %8, %2\n v_and_or_b32 %2, %2, %8, %3\n v_and_or_b32 %3, %3, %8, %4\n v_and_or_b32 %4, %4, %8, %5\n v_and_or_b32 %5, %5, %8, %6\n v_and_or_b32 %6, %6, %8, %7\n v_and_or_b32 %7, %7, %8, %0\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 11) { REP8(asm volatile("v_add3_u32 %0, %0, %8, %1\n v_add3_u32 %1, %1, %8, %2\n v_add3_u32 %2, %2, %8, %3\n v_add3_u32 %3, %3, %8, %4\n v_add3_u32 %4, %4, %8, %5\n v_add3_u32 %5, %5, %8, %6\n v_add3_u32 %6, %6, %8, %7\n v_add3_u32 %7, %7, %8, %0\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 12) { REP8(asm volatile("v_cmp_eq_u32 vcc, %0, %8\n v_cmp_eq_u32 vcc, %1, %8\n v_cmp_eq_u32 vcc, %2, %8\n v_cmp_eq_u32 vcc, %3, %8\n v_cmp_eq_u32 vcc, %4, %8\n v_cmp_eq_u32 vcc, %5, %8\n v_cmp_eq_u32 vcc, %6, %8\n v_cmp_eq_u32 vcc, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i) : "vcc");) }
        else if (KIND == 13) { REP8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i) : "vcc");) }
        else if (KIND == 14) { REP8(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %1, %2, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %3, %4, vcc\n v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %5, %6, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %6, %7, %0, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i) : "vcc");) }
        else if (KIND == 15) { REP8(asm volatile("v_bcnt_u32_b32 %0, %0, %8\n v_bcnt_u32_b32 %1, %1, %8\n v_bcnt_u32_b32 %2, %2, %8\n v_bcnt_u32_b32 %3, %3, %8\n v_bcnt_u32_b32 %4, %4, %8\n v_bcnt_u32_b32 %5, %5, %8\n v_bcnt_u32_b32 %6, %6, %8\n v_bcnt_u32_b32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 16) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 17) { REP8(asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i));) }
        else if (KIND == 18) { REP8(asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n s_add_u32 %4, %4, 1\n s_add_u32 %5, %5, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");) }
        else if (KIND == 19) { REP8(asm volatile("s_andn2_b32 %0, %0, %1\n s_brev_b32 %1, %1\n s_or_b32 %2, %2, %3\n s_lshr_b32 %3, %3, 1\n s_andn2_b32 %4, %4, %5\n s_brev_b32 %5, %5\n s_or_b32 %6, %6, %7\n s_lshr_b32 %7, %7, 1\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");) }
        else if (KIND == 20) { REP8(asm volatile("v_add_u32 %0, %0, %16\n s_add_u32 %8, %8, 1\n v_add_u32 %1, %1, %16\n s_add_u32 %9, %9, 1\n v_add_u32 %2, %2, %16\n s_add_u32 %10, %10, 1\n v_add_u32 %3, %3, %16\n s_add_u32 %11, %11, 1\n v_add_u32 %4, %4, %16\n s_add_u32 %12, %12, 1\n v_add_u32 %5, %5, %16\n s_add_u32 %13, %13, 1\n v_add_u32 %6, %6, %16\n s_add_u32 %14, %14, 1\n v_add_u32 %7, %7, %16\n s_add_u32 %15, %15, 1\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : "v"(i) : "scc");) }
        else if (KIND == 21) { REP8(asm volatile("v_readlane_b32 %8, %0, 3\n v_readlane_b32 %9, %1, 3\n v_readlane_b32 %10, %2, 3\n v_readlane_b32 %11, %3, 3\n v_readlane_b32 %12, %4, 3\n v_readlane_b32 %13, %5, 3\n v_readlane_b32 %14, %6, 3\n v_readlane_b32 %15, %7, 3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7));) }
        else if (KIND == 22) { REP8(asm volatile("v_writelane_b32 %0, %8, 3\n v_writelane_b32 %1, %9, 3\n v_writelane_b32 %2, %10, 3\n v_writelane_b32 %3, %11, 3\n v_writelane_b32 %4, %12, 3\n v_writelane_b32 %5, %13, 3\n v_writelane_b32 %6, %14, 3\n v_writelane_b32 %7, %15, 3\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7));) }
    }
    a0 ^= s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7;
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (unsigned)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int KIND>
static void run(const char *name, int n_cu, double ghz, unsigned *d) {
    const int iters = 2000;
    printf("%-15s", name);
    for (int w = 1; w <= 8; w *= 2) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(n_cu * w), dim3(256), 0, 0, d, 10);
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<KIND>, dim3(n_cu * w), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * ghz * 1e9;
        printf("  %d waves/SIMD: %.2f cyc", w, cyc / (64.0 * iters * w));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("%s: %d CUs, %.2f GHz; cycles per wave64 instruction and SIMD (independent instructions)\n", p.gcnArchName, p.multiProcessorCount, ghz);
    unsigned *d;
    (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    run<0>("v_add_u32", p.multiProcessorCount, ghz, d);
    run<1>("v_fma_f32", p.multiProcessorCount, ghz, d);
    run<2>("v_mov_b32", p.multiProcessorCount, ghz, d);
    run<3>("v_and_b32", p.multiProcessorCount, ghz, d);
    run<4>("v_or_b32", p.multiProcessorCount, ghz, d);
    run<5>("v_xor_b32", p.multiProcessorCount, ghz, d);
    run<6>("v_lshlrev_b32", p.multiProcessorCount, ghz, d);
    run<7>("v_lshrrev_b32", p.multiProcessorCount, ghz, d);
    run<8>("v_lshl_add_u32", p.multiProcessorCount, ghz, d);
    run<9>("v_bfe_u32", p.multiProcessorCount, ghz, d);
    run<10>("v_and_or_b32", p.multiProcessorCount, ghz, d);
    run<11>("v_add3_u32", p.multiProcessorCount, ghz, d);
    run<12>("v_cmp_eq_u32", p.multiProcessorCount, ghz, d);
    run<13>("v_cndmask_b32", p.multiProcessorCount, ghz, d);
    run<14>("cmp+cndmask", p.multiProcessorCount, ghz, d);
    run<15>("v_bcnt_u32_b32", p.multiProcessorCount, ghz, d);
    run<18>("s_add_u32", p.multiProcessorCount, ghz, d);
    run<19>("s_andn2/brev/or/lshr", p.multiProcessorCount, ghz, d);
    run<20>("v_add + s_add pairs", p.multiProcessorCount, ghz, d);
    run<21>("v_readlane_b32", p.multiProcessorCount, ghz, d);
    run<22>("v_writelane_b32", p.multiProcessorCount, ghz, d);
    run<16>("v_mul_lo_u32", p.multiProcessorCount, ghz, d);
    run<17>("v_mad_u32_u24", p.multiProcessorCount, ghz, d);
    return 0;
}
