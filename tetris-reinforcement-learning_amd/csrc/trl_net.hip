// trl_net.hip -- fused residual tower of AlphaSame for 16 filters (architectures.py:27-57, 119-127):
//   conv5x5(1->16) -> B x [BN, ReLU, conv3x3, BN, ReLU, conv3x3, +x] -> BN, ReLU -> conv1x1(16->1) -> BN, ReLU
// one 256-thread workgroup per 40x10 board, every activation of the board resident on the CU.
//
// Mapping (gfx950):
//   * a 3x3 convolution over 16 channels is the GEMM  out[400 x 16] = im2col[400 x 144] . W[144 x 16];
//     it runs on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains, the 1e-5 logit bar needs fp32):
//     M-tile = 16 board positions, N = the 16 output channels, K = 4 input channels of one tap;
//   * the conv input lives in LDS as [42 x 12 padded positions][16 channels]; the A fragment of the
//     four K-steps of one tap is ONE ds_read_b128 (the K order inside a tap is permuted so that a
//     lane's four channels are contiguous; the weights are packed on the host in the same order);
//   * the layer's whole B operand (144 x 16 weights) sits in 36 VGPRs per lane;
//   * the 5x5 stem runs on the matrix cores as well (im2col, K = 25 padded to 28);
//   * the residual stream never leaves registers (each wave owns 6-7 M-tiles in MFMA C/D layout);
//   * eval-mode batch norms are folded on the host (nets.py prepare()): per-channel scale/shift
//     before the first conv of a block, a bias after it;
//   * ONE LDS activation buffer (32 KB, rewritten in place behind a barrier) -> three boards per CU,
//     3 waves per SIMD.
// Per board: 20 layers x 25 tiles x 36 MFMAs = 18 000 MFMAs = 36.9 MFLOP; HBM traffic is the
// 1.6 KB board in, 1.6 KB out and the 188 KB of packed weights (L2-resident, shared by all boards).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "trl_internal.h"
using namespace trl;

namespace {

#ifdef TRL_PROFILE
__device__ unsigned long long g_tprof[2 * 8 * 10];  // [workgroup 0 | 1500][wave][phase] cycles (scripts/tower*_phase_probe.py)
#define TPROF_T(v) unsigned long long v = __builtin_readcyclecounter()
#define TPROF_ADD(i, a, b) do { tprof[i] += (b) - (a); } while (0)
#else
#define TPROF_T(v)
#define TPROF_ADD(i, a, b)
#endif

#ifndef TRL_EXP
#define TRL_EXP 0          // scripts/wide_exp.py: timing experiments that leave out one ingredient of a kernel (wrong results)
#endif
#if TRL_EXP == 5
#define TRL_BLOCK_SYNC() do { } while (0)          // k_trunk16s without its barriers (timing experiment)
#else
#define TRL_BLOCK_SYNC() __syncthreads()
#endif
constexpr int F = 16;
constexpr int PW = 12, PH = 42;                 // padded activation plane (1-cell zero border)
constexpr int STRIDE = 16;                      // floats per position
constexpr int BUF_FLOATS = PW * PH * STRIDE;    // 8064
constexpr int BW = 14, BH = 44;                 // padded input board (2-cell border) for the 5x5 conv
constexpr int NT = 25;                          // M-tiles of 16 positions
constexpr int WBLOCK = 16 + 16 + 2304 + 16 + 2304;   // packed floats per residual block
constexpr int TRUNK_THREADS = 256;

using f32x4 = __attribute__((ext_vector_type(4))) float;

// one 3x3 conv over the tiles of this wave: acc[s] (+)= conv(buf)[tile s], s = 0..5 (tiles wave + 4 s), and this
// wave's quarter of the K range of tile 24 (25 tiles do not divide by 4 waves: the odd tile is split over the
// 36 K-steps, 9 per wave, and summed through LDS -- every wave then carries 6.25 tiles instead of 7/6/6/6)
template <int W>
__device__ __forceinline__ void partial24(const float *buf, const float (&w)[36], int b24, int g, f32x4 &acc) {
#pragma unroll
    for (int tap = (9 * W) / 4; tap <= (9 * W + 8) / 4; tap++) {
        const int off = ((tap / 3) * PW + (tap % 3)) * STRIDE + 4 * g;
        f32x4 a = *reinterpret_cast<const f32x4 *>(buf + b24 + off);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int ks = tap * 4 + j;
            if (ks >= 9 * W && ks < 9 * W + 9) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], w[ks], acc, 0, 0, 0);
        }
    }
}

template <bool INIT_BIAS>
__device__ __forceinline__ void conv3x3(const float *buf, const float *wp, float bias, const int (&base)[7], int wave,
                                        int g, int lane, f32x4 (&acc)[7]) {
    float w[36];
#pragma unroll
    for (int i = 0; i < 36; i++) w[i] = wp[i * 64 + lane];
#pragma unroll
    for (int s = 0; s < 6; s++) {
        if (INIT_BIAS) acc[s] = f32x4{bias, bias, bias, bias};
        else acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    acc[6] = f32x4{0.f, 0.f, 0.f, 0.f};
    // tiles in pairs: two independent accumulator chains hide the 40-cycle dependent MFMA latency
#pragma unroll
    for (int s = 0; s < 6; s += 2) {
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            const int off = ((tap / 3) * PW + (tap % 3)) * STRIDE + 4 * g;
            f32x4 a0 = *reinterpret_cast<const f32x4 *>(buf + base[s] + off);
            f32x4 a1 = *reinterpret_cast<const f32x4 *>(buf + base[s + 1] + off);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[j], w[tap * 4 + j], acc[s], 0, 0, 0);
                acc[s + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[j], w[tap * 4 + j], acc[s + 1], 0, 0, 0);
            }
        }
    }
    if (wave == 0) partial24<0>(buf, w, base[6], g, acc[6]);
    else if (wave == 1) partial24<1>(buf, w, base[6], g, acc[6]);
    else if (wave == 2) partial24<2>(buf, w, base[6], g, acc[6]);
    else partial24<3>(buf, w, base[6], g, acc[6]);
}

__global__ __launch_bounds__(TRUNK_THREADS, 3) void k_trunk16(const float *__restrict__ boards, const float *__restrict__ wts,
                                                              int n_blocks, float *__restrict__ out,
                                                              const uint8_t *__restrict__ need) {
    if (need && !need[blockIdx.x]) return;      // board whose tower output is already cached (trl_mcts_zcache)
    TPROF_T(tp_start);
    // wave priority: what is not an MFMA loop (prologue, activation writes, reductions) runs at high priority -- it is
    // short, and the sooner a wave is back in a conv the fuller the matrix pipe (the arbiter otherwise serves the oldest
    // wave first, so a new workgroup's prologue crawls behind its neighbours' MFMA streams)
    __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) float lds[BUF_FLOATS + BW * BH + 4 * 256];
    float *bufA = lds, *brd = lds + BUF_FLOATS, *part = lds + BUF_FLOATS + BW * BH;      // part: K-split partials of tile 24
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, co = lane & 15;
    const size_t board = blockIdx.x;

    for (int i = tid; i < (BUF_FLOATS + BW * BH + 4 * 256) / 4; i += TRUNK_THREADS)
        reinterpret_cast<f32x4 *>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = tid; i < 400; i += TRUNK_THREADS) brd[(i / 10 + 2) * BW + (i % 10) + 2] = boards[board * 400 + i];
    __syncthreads();

    // tiles of this wave: T = wave + 4 s.  A-operand row of a lane: position 16 T + (lane & 15);
    // C/D rows of a lane: positions 16 T + 4 g + r, r = 0..3, column = output channel `co`.
    // slot 6 is tile 24 for every wave: wave 0 owns its outputs (C/D layout), all four waves share its K range
    int base[7];                    // float offset of the 3x3 window's top-left in a padded buffer (A operand)
#pragma unroll
    for (int s = 0; s < 7; s++) {
        int T = (s < 6) ? wave + 4 * s : NT - 1;
        int p = 16 * T + (lane & 15);
        base[s] = ((p / 10) * PW + (p % 10)) * STRIDE;
    }

    // conv1: 5x5, one input channel, as an im2col GEMM on the matrix cores too: K = 25 taps padded to 7 K-steps of 4
    // (A[m][k] = board value under tap 4 step + k, B[k][n] = w1[n][tap], zero for the three padding taps); the C/D
    // layout is that of the residual stream.  (On the vector ALU this stage is 700 LDS reads + 700 FMAs per lane and,
    // starved by the neighbours' MFMA streams, took a fifth of a workgroup's lifetime.)
    f32x4 x[7];
    {
        float wB[7];
        int toff[7];
#pragma unroll
        for (int st = 0; st < 7; st++) {
            const int t = 4 * st + g, tc = t < 25 ? t : 24;
            wB[st] = t < 25 ? wts[co * 25 + tc] : 0.f;
            toff[st] = (tc / 5) * BW + (tc % 5);
        }
#pragma unroll
        for (int s = 0; s < 7; s++) {
            int T = wave + 4 * s; if (T >= NT) T = NT - 1;
            const int p = 16 * T + (lane & 15), pb = (p / 10) * BW + (p % 10);
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 7; st++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(brd[pb + toff[st]], wB[st], c, 0, 0, 0);
            x[s] = c;
        }
    }

    const float *wb = wts + 400;
    f32x4 acc[7];
#ifdef TRL_PROFILE
    unsigned long long tprof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TPROF_T(tp_begin);
    for (int b = 0; b < n_blocks; b++, wb += WBLOCK) {
        TPROF_T(tp0);
        const float s1 = wb[co], t1 = wb[16 + co], bias_a = wb[32 + 2304 + co];
        // H1 = relu(bn1(x)) -> buffer A (own positions)
#pragma unroll
        for (int s = 0; s < 7; s++) {
            int T = wave + 4 * s;
            if (T < NT) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int p = 16 * T + 4 * g + r;
                    bufA[((p / 10 + 1) * PW + (p % 10) + 1) * STRIDE + co] = fmaxf(fmaf(x[s][r], s1, t1), 0.f);
                }
            }
        }
        TPROF_T(tp1);
        __syncthreads();
        TPROF_T(tp2);
        __builtin_amdgcn_s_setprio(0);
        conv3x3<true>(bufA, wb + 32, bias_a, base, wave, g, lane, acc);
        __builtin_amdgcn_s_setprio(3);
        *reinterpret_cast<f32x4 *>(part + wave * 256 + lane * 4) = acc[6];
        TPROF_T(tp3);
        __syncthreads();                              // every wave has read its halo: the buffer can be overwritten
        TPROF_T(tp4);
        if (wave == 0) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(part + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 256 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 512 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 768 + lane * 4);
            acc[6] = t + f32x4{bias_a, bias_a, bias_a, bias_a};
        }
#pragma unroll
        for (int s = 0; s < 7; s++) {
            int T = wave + 4 * s;
            if (T < NT) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int p = 16 * T + 4 * g + r;
                    bufA[((p / 10 + 1) * PW + (p % 10) + 1) * STRIDE + co] = fmaxf(acc[s][r], 0.f);
                }
            }
        }
        TPROF_T(tp5);
        __syncthreads();
        TPROF_T(tp6);
        __builtin_amdgcn_s_setprio(0);
        conv3x3<false>(bufA, wb + 32 + 2304 + 16, 0.f, base, wave, g, lane, acc);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s = 0; s < 6; s++) x[s] += acc[s];
        *reinterpret_cast<f32x4 *>(part + wave * 256 + lane * 4) = acc[6];
        TPROF_T(tp7);
        __syncthreads();
        TPROF_T(tp8);
        TPROF_ADD(0, tp0, tp1); TPROF_ADD(1, tp1, tp2); TPROF_ADD(2, tp2, tp3); TPROF_ADD(3, tp3, tp4); TPROF_ADD(4, tp4, tp5);
        TPROF_ADD(5, tp5, tp6); TPROF_ADD(6, tp6, tp7); TPROF_ADD(7, tp7, tp8);
        if (wave == 0) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(part + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 256 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 512 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 768 + lane * 4);
            x[6] += t;
        }
    }

#ifdef TRL_PROFILE
    {
        TPROF_T(tp_end);
        tprof[8] = tp_end - tp_begin; tprof[9] = tp_begin - tp_start;
        if ((blockIdx.x == 0 || blockIdx.x == 1500) && lane == 0)
            for (int i = 0; i < 10; i++) g_tprof[((blockIdx.x ? 1 : 0) * 8 + wave) * 10 + i] = tprof[i];
    }
#endif
    // tail: relu(bn(x)) -> 1x1 conv over the 16 channels (= 16 lanes) -> folded bn -> relu
    const float s = wb[co], t = wb[16 + co], wk = wb[32 + co], bk = wb[48];
#pragma unroll
    for (int sidx = 0; sidx < 7; sidx++) {
        int T = wave + 4 * sidx;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float v = fmaxf(fmaf(x[sidx][r], s, t), 0.f) * wk;
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (co == 0 && T < NT) out[board * 400 + 16 * T + 4 * g + r] = fmaxf(v + bk, 0.f);
        }
    }
}


// ---- the same tower on the f16 matrix rate, at fp32 accuracy: k_trunk16s -------------------------------------------
// The 1e-5 logit bar forbids a plain f16 / bf16 tower, and the fp32 MFMA runs at 1/16 of the 16-bit matrix rate.  But a
// fp32 number IS two halves: a = hi + lo / 2^11 with hi = f16(a), lo = f16((a - hi) * 2^11), exact to 2^-22 (|a - hi| <=
// 2^-11 |a|, rounded again to 11 bits; the scale keeps lo a normal number, so nothing depends on f16 denormals).  With
// activations and weights both split,
//     a * w = hi_a hi_w + (lo_a hi_w + hi_a lo_w) / 2^11 + O(2^-22),
// three v_mfma_f32_16x16x32_f16 (products of halves are exact in fp32, accumulation is fp32) into ONE accumulator: the
// weights' hi half is scaled by 2^11 in registers, so hi_a (2^11 hi_w), lo_a hi_w and hi_a lo_w all carry the factor 2^11
// (lo = 2^11 (x - hi) on both sides) and the sum is scaled back once per conv.  Measured against the reference's logits
// (tests/golden/net.npz, net_deep.npz): 1.8e-7 max error, the fp32 MFMA tower has 2.1e-7 (a 3-product bf16 split: 3e-6
// to 1.6e-5, rejected; 6 bf16 products match fp32 but cost twice the MFMAs of this scheme).
//   * K = 144 = 9 taps x 16 channels in 5 K-steps of 32 (two taps each, the tenth half-step has zero weights): a
//     lane's activation fragment is 8 consecutive channels of one tap = ONE ds_read_b128 for the hi half and one for the
//     lo half (rows of [12 hi positions | 12 lo positions][16 channels] halves at a pitch of 26 positions, 34 KB);
//   * 15 MFMAs of 16 cycles per (tile, conv) instead of 36 of 32 cycles: 4.8x fewer matrix-pipe cycles;
//   * weights: 5 K-steps x {hi, lo} x 4 VGPRs per lane, packed on the host (nets.AlphaSame._pack_tower_split), streamed
//     from L2 one K-step ahead (conv_step).
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr int HST = 16;                          // halves per position
// Row pitch of the split planes, in positions: a row holds its 12 hi positions (10 + the zero border) at 0..11 and its 12
// lo positions at 13..24.  26 = 10 + 16: stepping from one board row to the next moves a tile's positions by a multiple of
// 8 positions = 256 bytes = all 64 banks, so the 16 positions of a tile -- which span two or three board rows -- fall on
// distinct banks like 16 consecutive positions do (ds_read_b128: 4 cycles).  Two separate planes of pitch 12 moved the
// positions after a row end by 64 bytes and every fragment read took two passes (8 cycles; scripts/lds_bank_sim.py) --
// with 4 KB of fragments per wave and K-step that, not the matrix pipe, set the pace of the convs.
constexpr int SPW = 26, LO_OFS = 13 * HST;
constexpr int HPLANE = SPW * PH * HST / 2;       // both halves of every position, in floats (8736)
constexpr int WCONV_S = 5 * 2 * 64 * 4;          // packed dwords per 3x3 conv
constexpr int WBLOCK_S = 16 + 16 + WCONV_S + 16 + WCONV_S;
constexpr float SPLIT = 2048.f, INV_SPLIT = 1.f / 2048.f;

using f16x4 = __attribute__((ext_vector_type(4))) _Float16;

// wavefronts of the split towers whose activations left the f16 range since the counter was last reset (0 = none did)
__device__ unsigned int g_split_overflow;
constexpr float SPLIT_ACT_MAX = 65504.f;         // largest finite f16
// (per wave: a board whose four or eight waves all overflow counts that often -- the callers only ask "zero or not"; the
// library's workgroup reduction __syncthreads_or is avoided: in k_step it returned 0 for a set predicate now and then)
__device__ __forceinline__ void report_split_range(float mx) {
    if (__any(mx >= SPLIT_ACT_MAX) && (threadIdx.x & 63) == 0) atomicAdd(&g_split_overflow, 1u);
}

// four values (the lane's four channels of one position) -> hi and lo halves, one 8-byte store each.
// lo = f16((v - hi) * 2^11) as ONE v_fma_mix per value: fma(hi, -2^11, 2^11 v) -- the f16 source is widened inside the
// instruction, the sum is exact (|v - hi| <= 2^-11 |v|, so 2^11 (v - hi) has at most 13 significant bits) and the result is
// rounded to f16 once, like the four instructions (widen, subtract, scale, narrow) it replaces
// (the compiler widens and narrows around a packed fp32 fma instead, so the two instructions are written out)
__device__ __forceinline__ unsigned lo_pair(unsigned hpair, float a2, float b2) {
    unsigned d;
    const float negs = -SPLIT;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(hpair), "v"(negs), "v"(a2));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hpair), "v"(negs), "v"(b2));
    return d;
}
// `mx`: running maximum of everything this lane has split (the values are ReLU outputs, >= 0): two v_max3_f32 per call.  A
// value of 65 520 or more has no f16 hi half (it rounds to infinity and the convolution that reads it returns inf / NaN,
// which the next ReLU may even turn into a plausible 0): the kernels report it through g_split_overflow instead of
// returning such numbers silently (trl_tower_overflow; nets.py re-runs the board on the fp32 kernels).
#ifndef TRL_RANGE_GUARD
#define TRL_RANGE_GUARD 1       // 0: timing experiment only (scripts/guard_ab.py) -- the kernels then return inf / NaN silently
#endif
__device__ __forceinline__ void store_split4(_Float16 *hi, _Float16 *lo, int idx, f32x4 v, float &mx) {
#if TRL_RANGE_GUARD
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
#endif
    const f16x4 h = __builtin_convertvector(v, f16x4);                  // 2 x v_cvt_pk_f16_f32 (round to nearest even)
    const f32x4 v2 = v * SPLIT;
    const uint2 hp = __builtin_bit_cast(uint2, h);
    const uint2 lp = make_uint2(lo_pair(hp.x, v2[0], v2[1]), lo_pair(hp.y, v2[2], v2[3]));
    *reinterpret_cast<uint2 *>(hi + idx) = hp;
    *reinterpret_cast<uint2 *>(lo + idx) = lp;
}

// am[s] += 2^11 conv of the wave's tiles (slots 0..5 = tiles wave + 4 s; slot 6 = this wave's K-steps of tile 24),
// TRANSPOSED: the weights are the matrix instruction's A operand and the activations its B operand, so a lane of the
// result holds four consecutive output CHANNELS (4 g + r) of ONE position (tile position lane & 15) -- the split
// activations of a tile are then two 8-byte LDS stores per lane instead of eight 2-byte ones.
//   * K-step outermost: the weights of step c (hi and lo halves, 8 registers, + the hi half scaled by 2^11) serve all
//     tiles of the wave and the next step's are already on their way from L2 (40 registers held a whole conv's weights
//     before);
//   * ONE accumulator per tile as in k_trunk_wide_s: bs hi_a, bh lo_a and bl hi_a all carry the factor 2^11;
//   * the fragments of the next pair of tiles are requested before the matrix instructions of the current one are issued.
// K-step c of a lane reads tap 2 c + (g >> 1), channels 8 (g & 1) .. + 7.  The tap's offset is A_c + (g >> 1) D_c positions
// with A = {0, 2, SPW + 1, 2 SPW, 2 SPW + 2} and D = {1, SPW - 2, 1, 1, 0}: the lane-dependent part is one add, the rest an
// immediate of the ds_read.
struct SBase { const _Float16 *p0; };
template <int C> __device__ __forceinline__ const _Float16 *sb_ptr(const SBase &b, int d1, int d10) {
    constexpr int A[5] = {0, 2, SPW + 1, 2 * SPW, 2 * SPW + 2};
    return (C == 1 ? b.p0 + d10 : (C == 4 ? b.p0 : b.p0 + d1)) + A[C] * HST;
}
template <int C> __device__ __forceinline__ void sb_load(const SBase &b, int d1, int d10, f16x8 &h, f16x8 &l) {
    const _Float16 *p = sb_ptr<C>(b, d1, d10);
    h = *reinterpret_cast<const f16x8 *>(p);
    l = *reinterpret_cast<const f16x8 *>(p + LO_OFS);
}

struct SW { f16x8 bs, bh, bl, nh, nl; };          // weights of the K-step in progress (bs = 2^11 bh) and of the next one

// step I = 3 c + pair: tiles 2 pair, 2 pair + 1 on K-step c
template <bool ILV, int I>
__device__ __forceinline__ void conv_step(const SBase (&sb)[7], int d1, int d10, const uint4 *wp, SW &w, int wave,
                                          f16x8 (&h0)[2], f16x8 (&h1)[2], f16x8 (&l0)[2], f16x8 (&l1)[2], f32x4 (&am)[7]) {
    constexpr int c = I / 3, s = 2 * (I % 3), cur = I & 1, nxt = cur ^ 1;
    if constexpr (I % 3 == 0) {
        w.bh = w.nh; w.bl = w.nl; w.bs = w.bh * (_Float16)SPLIT;
        if constexpr (c + 1 < 5) {
            w.nh = __builtin_bit_cast(f16x8, wp[(2 * c + 2) * 64]);
            w.nl = __builtin_bit_cast(f16x8, wp[(2 * c + 3) * 64]);
        }
    }
    // (ILV: one fragment request in the shadow of each of the first four matrix instructions -- the variant for one board
    // per CU, where four ds_read_b128 issued in a row held the wave's issue for ~25 cycles per step,
    // scripts/mfma16_pattern.hip; with several waves per SIMD the others cover a group of four reads)
    constexpr bool LD = I + 1 < 15 && TRL_EXP != 6;
    constexpr int sn = LD ? 2 * ((I + 1) % 3) : 0, cn = LD ? (I + 1) / 3 : 0;
    if constexpr (LD && !ILV) {
        sb_load<cn>(sb[sn], d1, d10, h0[nxt], l0[nxt]);
        sb_load<cn>(sb[sn + 1], d1, d10, h1[nxt], l1[nxt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    am[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bs, h0[cur], am[s], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LD && ILV) h0[nxt] = *reinterpret_cast<const f16x8 *>(sb_ptr<cn>(sb[sn], d1, d10));
    __builtin_amdgcn_sched_barrier(0);
    am[s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bs, h1[cur], am[s + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LD && ILV) h1[nxt] = *reinterpret_cast<const f16x8 *>(sb_ptr<cn>(sb[sn + 1], d1, d10));
    __builtin_amdgcn_sched_barrier(0);
    am[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bh, l0[cur], am[s], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LD && ILV) l0[nxt] = *reinterpret_cast<const f16x8 *>(sb_ptr<cn>(sb[sn], d1, d10) + LO_OFS);
    __builtin_amdgcn_sched_barrier(0);
    am[s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bh, l1[cur], am[s + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (LD && ILV) l1[nxt] = *reinterpret_cast<const f16x8 *>(sb_ptr<cn>(sb[sn + 1], d1, d10) + LO_OFS);
    __builtin_amdgcn_sched_barrier(0);
    am[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bl, h0[cur], am[s], 0, 0, 0);
    am[s + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bl, h1[cur], am[s + 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I % 3 == 2) {
        // tile 24 (25 tiles do not divide by 4 waves): its 5 K-steps are shared out 2 / 1 / 1 / 1 and summed through LDS
        if ((c < 2 ? 0 : c - 1) == wave) {
            f16x8 h, l;
            sb_load<c>(sb[6], d1, d10, h, l);
            am[6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bs, h, am[6], 0, 0, 0);
            am[6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bh, l, am[6], 0, 0, 0);
            am[6] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w.bl, h, am[6], 0, 0, 0);
        }
    }
    if constexpr (I + 1 < 15) conv_step<ILV, I + 1>(sb, d1, d10, wp, w, wave, h0, h1, l0, l1, am);
}

// am[0..5] = bias + conv (bias only when INIT_BIAS), am[6] = the wave's share of tile 24 (no bias).  wp = the conv's packed
// weights + lane; w.nh / w.nl = K-step 0's weights, requested by the caller before the writes and the barrier that
// precede the conv
template <bool INIT_BIAS, bool ILV>
__device__ __forceinline__ void conv3x3s(const SBase (&sb)[7], int d1, int d10, const uint4 *wp, SW &w, f32x4 bias, int wave,
                                         f32x4 (&am)[7]) {
    f16x8 h0[2], h1[2], l0[2], l1[2];
    sb_load<0>(sb[0], d1, d10, h0[0], l0[0]);
    sb_load<0>(sb[1], d1, d10, h1[0], l1[0]);
#pragma unroll
    for (int s = 0; s < 6; s++) am[s] = INIT_BIAS ? bias * SPLIT : f32x4{0.f, 0.f, 0.f, 0.f};
    am[6] = f32x4{0.f, 0.f, 0.f, 0.f};
    conv_step<ILV, 0>(sb, d1, d10, wp, w, wave, h0, h1, l0, l1, am);
#pragma unroll
    for (int s = 0; s < 7; s++) am[s] *= INV_SPLIT;
}

// LEGAL: the first job.n_slots workgroups of the launch run the search's legal-move job instead of a board
// (leaf_legal_block, trl_internal.h: two of their four waves, the others end at once); board b is workgroup n_slots + b
#ifndef TRL_SPEC_WALK
#define TRL_SPEC_WALK 1      // 0: the legal-move job's waves 2, 3 end at once; a T search that needs the walk starts it after its closure
#endif
struct NoLegalJob {};
template <int MINB, bool ILV, bool LEGAL = false>
__global__ __launch_bounds__(TRUNK_THREADS, MINB) void k_trunk16s(const float *__restrict__ boards, const float *__restrict__ wts,
                                                                  int n_blocks, float *__restrict__ out,
                                                                  const uint8_t *__restrict__ need,
                                                                  std::conditional_t<LEGAL, LeafLegalJob, NoLegalJob> job = {}) {
    int block_id = blockIdx.x;
    if constexpr (LEGAL) {
        if (block_id < job.n_slots) {
            __shared__ BlockShared lsh;
#if TRL_CLOSURE && TRL_SPEC_WALK
            __shared__ SpecShared lsp;          // all four waves: two closures, two speculative ordered walks (T slots)
            leaf_legal_block_spec(job, lsh, lsp, block_id, threadIdx.x);
#else
            if (threadIdx.x >= BLOCK) return;
            leaf_legal_block(job, lsh, block_id, threadIdx.x);
#endif
            return;
        }
        block_id -= job.n_slots;
    }
    if (need && !need[block_id]) return;
    TPROF_T(tp_start);
    __builtin_amdgcn_s_setprio(3);
    __shared__ __attribute__((aligned(16))) float lds[HPLANE + BW * BH + 4 * 256];
    _Float16 *hiP = reinterpret_cast<_Float16 *>(lds), *loP = hiP + LO_OFS;
    float *brd = lds + HPLANE, *part = lds + HPLANE + BW * BH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, pn = lane & 15;      // result layout: channels 4 g + r of tile position pn
    const size_t board = (size_t)block_id;

    for (int i = tid; i < (HPLANE + BW * BH + 4 * 256) / 4; i += TRUNK_THREADS)
        reinterpret_cast<f32x4 *>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = tid; i < 400; i += TRUNK_THREADS) brd[(i / 10 + 2) * BW + (i % 10) + 2] = boards[board * 400 + i];
    __syncthreads();

    SBase sb[7];                    // the lane's position in tile s: base of its 3x3 window in the hi plane (see SBase)
#pragma unroll
    for (int s = 0; s < 7; s++) {
        const int T = (s < 6) ? wave + 4 * s : NT - 1;
        const int p = 16 * T + pn;
        sb[s].p0 = hiP + ((p / 10) * SPW + (p % 10)) * HST + 8 * (g & 1);
    }
    const int d1 = (g >> 1) * HST, d10 = (g >> 1) * (SPW - 2) * HST;
    // where the lane's four channels of its position go in the planes: sb[s].p0 + WOFS (window centre, channels 4 g ..)
    const int WOFS = (SPW + 1) * HST + 4 * g - 8 * (g & 1);
    // 5x5 stem, one input channel: im2col GEMM on the fp32 matrix instruction as in k_trunk16 (7 per tile, once), with the
    // operands swapped like the convs': A = w1[channel = lane & 15][tap 4 st + g], B = board value under that tap at the
    // lane's position
    f32x4 x[7];
    {
        float wB[7];
        int toff[7];
#pragma unroll
        for (int st = 0; st < 7; st++) {
            const int t = 4 * st + g, tc = t < 25 ? t : 24;
            wB[st] = t < 25 ? wts[pn * 25 + tc] : 0.f;
            toff[st] = (tc / 5) * BW + (tc % 5);
        }
#pragma unroll
        for (int s = 0; s < 7; s++) {
            int T = wave + 4 * s; if (T >= NT) T = NT - 1;
            const int p = 16 * T + pn, pb = (p / 10) * BW + (p % 10);
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 7; st++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[st], brd[pb + toff[st]], c, 0, 0, 0);
            x[s] = c;
        }
    }

    const float *wb = wts + 400;
    f32x4 am[7];
    SW w;
    float act_max = 0.f;            // largest activation this lane has split (store_split4)
#ifdef TRL_PROFILE
    unsigned long long tprof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TPROF_T(tp_begin);
    for (int b = 0; b < n_blocks; b++, wb += WBLOCK_S) {
        TPROF_T(tp0);
        const f32x4 s1 = *reinterpret_cast<const f32x4 *>(wb + 4 * g), t1 = *reinterpret_cast<const f32x4 *>(wb + 16 + 4 * g);
        const f32x4 bias_a = *reinterpret_cast<const f32x4 *>(wb + 32 + WCONV_S + 4 * g);
        const uint4 *wpa = reinterpret_cast<const uint4 *>(wb + 32) + lane, *wpb = reinterpret_cast<const uint4 *>(wb + 32 + WCONV_S + 16) + lane;
        w.nh = __builtin_bit_cast(f16x8, wpa[0]); w.nl = __builtin_bit_cast(f16x8, wpa[64]);
#pragma unroll
        for (int s = 0; s < 7; s++) {                               // H1 = relu(bn1(x)), split, own positions
            if ((s < 6 || wave == 0) && (TRL_EXP != 7 || s == 0)) {
                f32x4 v = __builtin_elementwise_fma(x[s], s1, t1);
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                store_split4(const_cast<_Float16 *>(sb[s].p0), const_cast<_Float16 *>(sb[s].p0) + LO_OFS, WOFS, v, act_max);
            }
        }
        TPROF_T(tp1);
        TRL_BLOCK_SYNC();
        TPROF_T(tp2);
        __builtin_amdgcn_s_setprio(0);
        conv3x3s<true, ILV>(sb, d1, d10, wpa, w, bias_a, wave, am);
        __builtin_amdgcn_s_setprio(3);
        *reinterpret_cast<f32x4 *>(part + wave * 256 + lane * 4) = am[6];
        w.nh = __builtin_bit_cast(f16x8, wpb[0]); w.nl = __builtin_bit_cast(f16x8, wpb[64]);      // conv b's first weights
        TPROF_T(tp3);
        TRL_BLOCK_SYNC();                              // every wave has read its halo: the planes can be overwritten
        TPROF_T(tp4);
        if (wave == 0) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(part + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 256 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 512 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 768 + lane * 4);
            am[6] = t + bias_a;
        }
#pragma unroll
        for (int s = 0; s < 7; s++) {
            if ((s < 6 || wave == 0) && (TRL_EXP != 7 || s == 0)) {
                f32x4 v = am[s];
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                store_split4(const_cast<_Float16 *>(sb[s].p0), const_cast<_Float16 *>(sb[s].p0) + LO_OFS, WOFS, v, act_max);
            }
        }
        TPROF_T(tp5);
        TRL_BLOCK_SYNC();
        TPROF_T(tp6);
        __builtin_amdgcn_s_setprio(0);
        conv3x3s<false, ILV>(sb, d1, d10, wpb, w, f32x4{0.f, 0.f, 0.f, 0.f}, wave, am);
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s = 0; s < 6; s++) x[s] += am[s];
        *reinterpret_cast<f32x4 *>(part + wave * 256 + lane * 4) = am[6];
        TPROF_T(tp7);
        TRL_BLOCK_SYNC();
        TPROF_T(tp8);
        TPROF_ADD(0, tp0, tp1); TPROF_ADD(1, tp1, tp2); TPROF_ADD(2, tp2, tp3); TPROF_ADD(3, tp3, tp4); TPROF_ADD(4, tp4, tp5);
        TPROF_ADD(5, tp5, tp6); TPROF_ADD(6, tp6, tp7); TPROF_ADD(7, tp7, tp8);
        if (wave == 0) {
            f32x4 t = *reinterpret_cast<const f32x4 *>(part + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 256 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 512 + lane * 4);
            t += *reinterpret_cast<const f32x4 *>(part + 768 + lane * 4);
            x[6] += t;
        }
    }
#ifdef TRL_PROFILE
    {
        TPROF_T(tp_end);
        tprof[8] = tp_end - tp_begin; tprof[9] = tp_begin - tp_start;
        if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0)
            for (int i = 0; i < 10; i++) g_tprof[((blockIdx.x ? 1 : 0) * 8 + wave) * 10 + i] = tprof[i];
    }
#endif
    report_split_range(act_max);
    // tail: relu(bn(x)) -> 1x1 conv over the 16 channels (4 in the lane, 4 lane groups) -> folded bn -> relu   (fp32)
    const f32x4 s = *reinterpret_cast<const f32x4 *>(wb + 4 * g), t = *reinterpret_cast<const f32x4 *>(wb + 16 + 4 * g);
    const f32x4 wk = *reinterpret_cast<const f32x4 *>(wb + 32 + 4 * g);
    const float bk = wb[48];
#pragma unroll
    for (int sidx = 0; sidx < 7; sidx++) {
        const int T = wave + 4 * sidx;
        f32x4 y = x[sidx] * s + t;
        float v = fmaxf(y[0], 0.f) * wk[0];
        v = fmaf(fmaxf(y[1], 0.f), wk[1], v); v = fmaf(fmaxf(y[2], 0.f), wk[2], v); v = fmaf(fmaxf(y[3], 0.f), wk[3], v);
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (g == 0 && T < NT) out[board * 400 + 16 * T + pn] = fmaxf(v + bk, 0.f);
    }
}


// ---- wide tower: 32 / 64 filters (BASELINE config C5: 20 blocks x 64 filters) ------------------------------------
// Same algorithm and data layout as k_trunk16, generalised over the channel count FF (a multiple of 16):
//   * the activations are FF/4 quad-planes of [42 x 12 padded positions][4 channels] in LDS (FF = 64: 126 KB, one
//     board per CU; FF = 32: 63 KB, two boards per CU); the A fragment of (tap, 16-channel group) is one
//     conflict-free ds_read_b128 per lane (lane (m, g) reads quad g of position m);
//   * 2 NN waves (NN = FF/16 N-tiles): wave = ntile + NN * mq owns the 16 output channels of N-tile `ntile` for the
//     even (mq = 0: 13) or odd (mq = 1: 12) M-tiles; FF = 64 -> 8 waves, 25 tiles per SIMD; two waves per SIMD
//     either way (256 VGPRs each);
//   * a conv's B operand is 9 FF x FF floats (147 KB at FF = 64): too large for registers, so it streams from
//     L2 one (tap, channel group) step ahead of the MFMAs -- 4 VGPRs per step, reused by every M-tile of the wave
//     (52 MFMAs per 1 KB of weights);
//   * residual stream and accumulators stay in registers in MFMA C/D layout (2 x 52 VGPRs at FF = 64).
// Per board at FF = 64, 20 blocks: 40 convs x 100 (M,N) tiles x 144 K-steps = 576 000 MFMAs = 1.18 GFLOP.
template <int FF> struct Wide {
    static constexpr int NN = FF / 16, CG = FF / 16, WAVES = 2 * NN, MSPLIT = 2, TS = (NT + MSPLIT - 1) / MSPLIT;
    static constexpr int THREADS = 64 * WAVES;
    // LDS layout of a 16-channel group: 4 quad-planes [42 x 12 padded positions][4 channels] (+4 floats of padding per
    // quad-plane, which spreads the C/D-layout ds_write_b32 of the 4 channel quads over the banks).  Measured: planes
    // padded to 2048 floats (A reads conflict-free by the 4 x 16-lane ds_read_b128 bank rule) ran 1.5 % slower, and the
    // [position][16 channels] layout of the 16-filter kernel the same as this one -- LDS is 1/3 busy, not the limit.
    static constexpr int QS = 4, PLANEQ = PW * PH * QS + 4, PLANE = 4 * PLANEQ;
    static constexpr int LDS_FLOATS = CG * PLANE + BW * BH + 200;   // planes, padded input board, write-offset table
    static constexpr int CONVW = 9 * FF * FF;
    static constexpr int WBLOCKW = 3 * FF + 2 * CONVW;
};

template <int FF, int S0, int S1>
__device__ __forceinline__ void wide_load(const float *pl, const int (&base)[Wide<FF>::TS], int off, f32x4 (&a)[S1 - S0]) {
#pragma unroll
    for (int s = S0; s < S1; s++) a[s - S0] = *reinterpret_cast<const f32x4 *>(pl + base[s] + off);
}

template <int FF, int S0, int S1>
__device__ __forceinline__ void wide_mfma(const f32x4 (&a)[S1 - S0], const float (&w)[4], f32x4 (&acc)[Wide<FF>::TS]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {           // S1 - S0 independent accumulator chains per K-step
#pragma unroll
        for (int s = S0; s < S1; s++) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s - S0][j], w[j], acc[s], 0, 0, 0);
    }
}

// acc[0 .. NTL) += conv3x3(planes)[tiles of this wave]; NTL = 13 (even tiles) or 12 (odd tiles).
// wn = the first K-step's B operands, already in flight (loaded before the barrier that precedes the conv).
// Software pipeline over the 36 (channel group, tap) steps x 2 tile groups: the A fragments of the next group are
// requested before the MFMAs of the current one are issued, so a wave running alone on its SIMD does not stall on LDS.
template <int FF, int NTL>
__device__ __forceinline__ void conv_wide(const float *lds, const float *wq, float (&wn)[4], const int (&base)[Wide<FF>::TS],
                                          int g, f32x4 (&acc)[Wide<FF>::TS]) {
    using W = Wide<FF>;
    constexpr int WSTEP = W::NN * 64, H = (NTL + 1) / 2;
    f32x4 a0[H], a1[NTL - H];
    const float *pl = lds + g * W::PLANEQ;
    wide_load<FF, 0, H>(pl, base, 0, a0);
    for (int cg = 0; cg < W::CG; cg++) {
        const float *pn = lds + (cg + 1 < W::CG ? cg + 1 : cg) * W::PLANE + g * W::PLANEQ;   // (the last prefetch of a conv is unused)
#pragma unroll
        for (int tap = 0; tap < 9; tap++) {
            float w[4];
#pragma unroll
            for (int j = 0; j < 4; j++) w[j] = wn[j];
            wq += 4 * WSTEP;                          // next (tap, channel group); the last prefetch of a conv reads the
#pragma unroll                                        // first words of what follows it in the packed buffer (unused)
            for (int j = 0; j < 4; j++) wn[j] = wq[j * WSTEP];
            const int off = ((tap / 3) * PW + (tap % 3)) * W::QS;
            const int offn = (((tap + 1) % 9 / 3) * PW + ((tap + 1) % 9 % 3)) * W::QS;
            wide_load<FF, H, NTL>(pl, base, off, a1);
            __builtin_amdgcn_sched_barrier(0);        // keep the weight / A-fragment requests ahead of the MFMAs that hide them
            wide_mfma<FF, 0, H>(a0, w, acc);
            __builtin_amdgcn_sched_barrier(0);
            wide_load<FF, 0, H>(tap == 8 ? pn : pl, base, offn, a0);
            __builtin_amdgcn_sched_barrier(0);
            wide_mfma<FF, H, NTL>(a1, w, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        pl = pn;
    }
}

// POST = false: AlphaSame (pre-activation blocks, 5x5 stem, tail BN-ReLU-1x1-BN-ReLU -> out[board][400])
// POST = true : BaseResNet family (architectures.py:159-170, 215-222: 3x3 stem + BN + ReLU, post-activation blocks
//               x = relu(bn2(conv2(relu(bn1(conv1(x))))) + x); tail = kout raw 1x1 projections -> out[board][kout][400],
//               their bias / conditioning / ReLU are applied by the caller)
template <int FF, bool POST>
__global__ __launch_bounds__(Wide<FF>::THREADS, 2) void k_trunk_wide(const float *__restrict__ boards,
                                                                     const float *__restrict__ wts, int n_blocks,
                                                                     float *__restrict__ out,
                                                                     const uint8_t *__restrict__ need, int kout) {
    using W = Wide<FF>;
    constexpr int TS = W::TS, NN = W::NN, MSPLIT = W::MSPLIT, THREADS = W::THREADS, WSTEP = W::NN * 64;
    constexpr int STEM = POST ? 9 : 25;
    if (need && !need[blockIdx.x]) return;
    __shared__ __attribute__((aligned(16))) float lds[W::LDS_FLOATS];
    float *brd = lds + W::CG * W::PLANE;
    // otab[p / 4] = the plane offsets (in floats) of positions p .. p + 3, four u16: the write phases look them up
    // instead of dividing by the board width
    uint2 *otab = reinterpret_cast<uint2 *>(brd + BW * BH);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, co = lane & 15;
    const int ntile = wave % NN, mq = wave / NN;
    const int ch = ntile * 16 + co;
    const size_t board = blockIdx.x;

    for (int i = tid; i < W::LDS_FLOATS; i += THREADS) lds[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 400; i += THREADS) brd[(i / 10 + 2) * BW + (i % 10) + 2] = boards[board * 400 + i];
    for (int i = tid; i < 100; i += THREADS) {
        unsigned o[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { int p = 4 * i + r; o[r] = ((p / 10 + 1) * PW + (p % 10) + 1) * W::QS; }
        otab[i] = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    }
    __syncthreads();

    // slot s of this wave is M-tile mq + 2 s (slot 12 of the odd waves does not exist: clamped here, never computed)
    int base[TS];
#pragma unroll
    for (int s = 0; s < TS; s++) {
        int T = mq + MSPLIT * s; if (T >= NT) T = NT - 1;
        int p = 16 * T + (lane & 15);
        base[s] = ((p / 10) * PW + (p % 10)) * W::QS;
    }

    f32x4 x[TS];
    {
        float w1[STEM];
#pragma unroll
        for (int i = 0; i < STEM; i++) w1[i] = wts[ch * STEM + i];
        const float b0 = POST ? wts[FF * STEM + ch] : 0.f;
#pragma unroll
        for (int s = 0; s < TS; s++) {
            int T = mq + MSPLIT * s; if (T >= NT) T = NT - 1;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int p = 16 * T + 4 * g + r, py = p / 10, px = p % 10;
                float a = b0;
                if (POST) {
#pragma unroll
                    for (int k = 0; k < 9; k++) a = fmaf(w1[k], brd[(py + 1 + k / 3) * BW + px + 1 + (k % 3)], a);
                    a = fmaxf(a, 0.f);
                } else {
#pragma unroll
                    for (int k = 0; k < 25; k++) a = fmaf(w1[k], brd[(py + k / 5) * BW + px + (k % 5)], a);
                }
                x[s][r] = a;
            }
        }
    }

    float *own = lds + ntile * W::PLANE + (co >> 2) * W::PLANEQ + (co & 3);   // this lane's output channel in its quad-plane
    const float *wb = wts + FF * STEM + (POST ? FF : 0);
    f32x4 acc[TS];
#ifdef TRL_PROFILE
    unsigned long long tprof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TPROF_T(tp_begin);
    // two layers per residual block, one loop body for both (the conv code exists once per tile count)
    for (int layer = 0; layer < 2 * n_blocks; layer++) {
        const bool first = !(layer & 1);
        float sc = 1.f, sh = 0.f, bias = 0.f;
        const float *wq;
        if (POST) {                                    // block: b1[F] W1 b2[F] W2
            bias = first ? wb[ch] : wb[FF + W::CONVW + ch];
            wq = wb + (first ? FF : 2 * FF + W::CONVW);
        } else {                                       // block: s1[F] t1[F] Wa ba[F] Wb
            if (first) { sc = wb[ch]; sh = wb[FF + ch]; bias = wb[2 * FF + W::CONVW + ch]; }
            wq = wb + (first ? 2 * FF : 3 * FF + W::CONVW);
        }
        wq += ntile * 64 + lane;
        float wn[4];
        TPROF_T(tp0);
#pragma unroll
        for (int j = 0; j < 4; j++) wn[j] = wq[j * WSTEP];
        // conv input -> planes (own positions): the block input (pre-activation: through bn1 + relu) for the first
        // conv, relu(previous conv) for the second
#pragma unroll
        for (int s = 0; s < (TRL_EXP == 3 ? 1 : TS); s++) {
            int T = mq + MSPLIT * s;
            if (T < NT) {
                const uint2 o = otab[4 * T + g];
                const unsigned off[4] = {o.x & 0xffffu, o.x >> 16, o.y & 0xffffu, o.y >> 16};
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = first ? (POST ? x[s][r] : fmaf(x[s][r], sc, sh)) : acc[s][r];
                    own[off[r]] = fmaxf(v, 0.f);
                }
            }
        }
        TPROF_T(tp1);
        __syncthreads();
        TPROF_T(tp2);
#pragma unroll
        for (int s = 0; s < TS; s++) acc[s] = f32x4{bias, bias, bias, bias};
        if (mq == 0) conv_wide<FF, TS>(lds, wq, wn, base, g, acc);
        else conv_wide<FF, NT - TS>(lds, wq, wn, base, g, acc);
        TPROF_T(tp3);
        if (!first) {
#pragma unroll
            for (int s = 0; s < TS; s++) {
                x[s] += acc[s];
                if (POST) { x[s][0] = fmaxf(x[s][0], 0.f); x[s][1] = fmaxf(x[s][1], 0.f); x[s][2] = fmaxf(x[s][2], 0.f); x[s][3] = fmaxf(x[s][3], 0.f); }
            }
            wb += POST ? 2 * FF + 2 * W::CONVW : W::WBLOCKW;
        }
        TPROF_T(tp4);
        __syncthreads();                              // every wave has read its inputs: the planes can be overwritten
        TPROF_T(tp5);
        TPROF_ADD(0, tp0, tp1); TPROF_ADD(1, tp1, tp2); TPROF_ADD(2, tp2, tp3); TPROF_ADD(3, tp3, tp4); TPROF_ADD(4, tp4, tp5);
    }
#ifdef TRL_PROFILE
    {
        TPROF_T(tp_end);
        tprof[5] = tp_end - tp_begin;
        if ((blockIdx.x == 0 || blockIdx.x == 1500) && lane == 0)
            for (int i = 0; i < 10; i++) g_tprof[((blockIdx.x ? 1 : 0) * 8 + wave) * 10 + i] = tprof[i];
    }
#endif

    float *part = lds;                                 // [kout][NN][400] partial sums (the planes are dead now)
    if (POST) {
        // tail: kout 1x1 projections of the feature map, raw
        for (int k = 0; k < kout; k++) {
            const float wk = wb[k * FF + ch];
#pragma unroll
            for (int sidx = 0; sidx < TS; sidx++) {
                int T = mq + MSPLIT * sidx;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float v = x[sidx][r] * wk;
                    v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                    if (co == 0 && T < NT) part[(k * NN + ntile) * 400 + 16 * T + 4 * g + r] = v;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < 400 * kout; i += THREADS) {
            const int k = i / 400, pos = i % 400;
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NN; n++) v += part[(k * NN + n) * 400 + pos];
            out[(board * kout + k) * 400 + pos] = v;
        }
    } else {
        // tail: relu(bn(x)) -> 1x1 conv over the FF channels (16 lanes x NN waves) -> folded bn -> relu
        const float s = wb[ch], t = wb[FF + ch], wk = wb[2 * FF + ch], bk = wb[3 * FF];
#pragma unroll
        for (int sidx = 0; sidx < TS; sidx++) {
            int T = mq + MSPLIT * sidx;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float v = fmaxf(fmaf(x[sidx][r], s, t), 0.f) * wk;
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
                if (co == 0 && T < NT) part[ntile * 400 + 16 * T + 4 * g + r] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < 400; i += THREADS) {
            float v = bk;
#pragma unroll
            for (int n = 0; n < NN; n++) v += part[n * 400 + i];
            out[board * 400 + i] = fmaxf(v, 0.f);
        }
    }
}

// ---- wide towers on the f16 matrix rate with split operands (see k_trunk16s): k_trunk_wide_s<32|64> -----------------
// Same scheme as k_trunk16s -- a = hi + lo / 2^11, three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation -- on the
// data flow of k_trunk_wide (weights streamed from L2 one K-step ahead and reused by all tiles of the wave; residual
// stream and accumulators in registers).  K = 9 taps x FF channels in 9 FF/32 K-steps of 32 channels of one tap (no
// padding: 576 = 18 x 32): 54 matrix instructions of 16 cycles per (tile, channel group, conv) at FF = 64 instead of 144
// of 32 cycles.
//   * 2 FF/16 waves: wave = cgroup + NN * mq owns the 16 output channels of group `cgroup` for the even (mq = 0: 13) or odd
//     (mq = 1: 12) position tiles -- two waves per SIMD;
//   * the result is TRANSPOSED as in k_trunk16s (weights = the instruction's A operand, activations = B): a lane holds
//     channels 4 g + r of its group at tile position lane & 15, so a tile's split activations leave as two 8-byte LDS
//     stores per lane (104 two-byte stores per lane and layer before: that phase, not the convs, was the largest single
//     cost of the layer -- a quarter of the launch in a leave-one-out timing, scripts/wide_exp.py);
//   * ONE accumulator per tile: with the weights' hi half scaled by 2^11 in registers (bs = 2048 bh, exact; |w| < 32 is
//     checked by the packer) the three products bs hi_a, bh lo_a and bl hi_a all carry the factor 2^11 (lo = 2^11 (a - hi)
//     on both sides), so the correction needs no accumulator of its own -- 52 registers, which hold a second fragment
//     buffer instead: the fragments of the next group of 2-3 tiles are requested before the 6-9 matrix instructions of
//     the current one (phase stamps showed a wave spending half of a conv waiting for the reads it had just issued);
//   * activations: FF/8 "oct-planes" of halves, [42 rows][12 hi | 12 lo positions at a pitch of 26][8 channels] (17.25 KB):
//     the fragment of a lane (position n, channels 8 g .. 8 g + 7 of the K-step's 32) is ONE ds_read_b128 from plane
//     4 kg + g, the 16 positions of a tile on distinct banks although they span board rows (pitch 26 = 10 + 16, see SPW;
//     separate hi / lo planes at pitch 12 took two LDS passes per read); FF = 64: 138 KB, one board per CU;
//   * packed weights (nets._pack_conv3x3_split_wide): [K-step = kg * 9 + tap][hi | lo][channel group][lane][8 halves], lane
//     (m, g) holds W[16 cgroup + m][32 kg + 8 g + j][tap]; 147 KB per conv at FF = 64, as many bytes as the fp32 layout.
template <int FF> struct WideS {
    static constexpr int NN = FF / 16, KG = FF / 32, WAVES = 2 * NN, TS = 13;
    static constexpr int THREADS = 64 * WAVES, OCT = FF / 8;
    // halves per oct-plane: 42 rows x SPW positions x 8 channels (hi at positions 0..11 of a row, lo at 13..24, as in
    // k_trunk16s), padded to a multiple of 256 bytes: lanes 0-3 / 12-15 (plane g = 0) and 20-27 (plane g = 1) are served
    // together, so the planes must sit a whole number of bank rounds apart for their cells to interleave
    static constexpr int PLANEH = 8832, LO = 13 * 8;
    static constexpr int PLANES_FLOATS = OCT * PLANEH / 2;
    static constexpr int LDS_FLOATS = PLANES_FLOATS + BW * BH;
    static constexpr int KSTEPS = 9 * KG;
    static constexpr int WSTEP = 2 * NN * 64 * 4;          // packed dwords per K-step (hi and lo, all channel groups)
    static constexpr int CONVW = KSTEPS * WSTEP;
};

// tiles of group G (of 6) of a wave with NTL tiles: 2, 2, 2, 2, 2 and 2 or 3
template <int NTL, int G> struct SGrp { static constexpr int S0 = 2 * G, S1 = (G == 5) ? NTL : 2 * G + 2, N = S1 - S0; };

template <int NTL, int G, int NB>
__device__ __forceinline__ void wides_load(const _Float16 *ph, const _Float16 *pl, const int (&base)[13], int off,
                                           f16x8 (&ah)[NB], f16x8 (&al)[NB]) {
#pragma unroll
    for (int s = SGrp<NTL, G>::S0; s < SGrp<NTL, G>::S1; s++) {
        ah[s - SGrp<NTL, G>::S0] = *reinterpret_cast<const f16x8 *>(ph + base[s] + off);
        al[s - SGrp<NTL, G>::S0] = *reinterpret_cast<const f16x8 *>(pl + base[s] + off);
    }
}
template <int NTL, int G, int NB>
__device__ __forceinline__ void wides_mfma(const f16x8 (&ah)[NB], const f16x8 (&al)[NB], f16x8 bs, f16x8 bh, f16x8 bl, f32x4 (&am)[13]) {
    constexpr int S0 = SGrp<NTL, G>::S0, S1 = SGrp<NTL, G>::S1;
#pragma unroll
    for (int s = S0; s < S1; s++) am[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bs, ah[s - S0], am[s], 0, 0, 0);
#pragma unroll
    for (int s = S0; s < S1; s++) am[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[s - S0], am[s], 0, 0, 0);
#pragma unroll
    for (int s = S0; s < S1; s++) am[s] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[s - S0], am[s], 0, 0, 0);
}

// am[s] += 2^11 conv of the wave's NTL tiles; wh / wl = the first K-step's weights (already in flight)
template <int FF, int NTL>
__device__ __forceinline__ void conv_wide_s(const _Float16 *hiL, const _Float16 *loL, const uint4 *wq, f16x8 wh, f16x8 wl,
                                            const int (&base)[13], f32x4 (&am)[13]) {
    using W = WideS<FF>;
    constexpr int STEPQ = W::WSTEP / 4;
    f16x8 ah0[2], al0[2], ah1[3], al1[3];
    wides_load<NTL, 0>(hiL, loL, base, 0, ah0, al0);
#if TRL_EXP == 2
    wides_load<NTL, 5>(hiL, loL, base, 0, ah1, al1);
#endif
    for (int kg = 0; kg < W::KG; kg++) {
        const _Float16 *ph = hiL + kg * 4 * W::PLANEH, *pl = loL + kg * 4 * W::PLANEH;
#pragma unroll 1
        for (int tap = 0; tap < 9; tap++) {
            const f16x8 bh = wh, bl = wl, bs = bh * (_Float16)SPLIT;
            wq += STEPQ;                                   // next K-step (the last prefetch of a conv reads what follows it: unused)
#if TRL_EXP != 1
            wh = __builtin_bit_cast(f16x8, wq[0]);
            wl = __builtin_bit_cast(f16x8, wq[W::NN * 64]);
#endif
            const int ty = tap / 3, off = (ty * SPW + (tap - 3 * ty)) * 8;
            const int tn = tap == 8 ? 0 : tap + 1, tny = tn / 3, offn = (tny * SPW + (tn - 3 * tny)) * 8;
            const _Float16 *phn = (tap == 8 && kg + 1 < W::KG) ? ph + 4 * W::PLANEH : ph;      // (the last prefetch of a conv is unused)
            const _Float16 *pln = (tap == 8 && kg + 1 < W::KG) ? pl + 4 * W::PLANEH : pl;
#if TRL_EXP == 2
            wides_mfma<NTL, 0>(ah0, al0, bs, bh, bl, am); wides_mfma<NTL, 1>(ah0, al0, bs, bh, bl, am);
            wides_mfma<NTL, 2>(ah0, al0, bs, bh, bl, am); wides_mfma<NTL, 3>(ah0, al0, bs, bh, bl, am);
            wides_mfma<NTL, 4>(ah0, al0, bs, bh, bl, am); wides_mfma<NTL, 5>(ah1, al1, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
            continue;
#endif
            wides_load<NTL, 1>(ph, pl, base, off, ah1, al1);
            __builtin_amdgcn_sched_barrier(0);
            wides_mfma<NTL, 0>(ah0, al0, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
            wides_load<NTL, 2>(ph, pl, base, off, ah0, al0);
            __builtin_amdgcn_sched_barrier(0);
            wides_mfma<NTL, 1>(ah1, al1, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
            wides_load<NTL, 3>(ph, pl, base, off, ah1, al1);
            __builtin_amdgcn_sched_barrier(0);
            wides_mfma<NTL, 2>(ah0, al0, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
            wides_load<NTL, 4>(ph, pl, base, off, ah0, al0);
            __builtin_amdgcn_sched_barrier(0);
            wides_mfma<NTL, 3>(ah1, al1, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
            wides_load<NTL, 5>(ph, pl, base, off, ah1, al1);
            __builtin_amdgcn_sched_barrier(0);
            wides_mfma<NTL, 4>(ah0, al0, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
            wides_load<NTL, 0>(phn, pln, base, offn, ah0, al0);
            __builtin_amdgcn_sched_barrier(0);
            wides_mfma<NTL, 5>(ah1, al1, bs, bh, bl, am);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    return v;
}

// one board through the tower; all threads of the workgroup call it with the same board (`lds`: W::LDS_FLOATS floats)
template <int FF, bool POST>
static __device__ __forceinline__ void trunk_wide_s_board(const float *__restrict__ boards, const float *__restrict__ wts, int n_blocks,
                                                          float *__restrict__ out, int kout, const size_t board, float *lds) {
    using W = WideS<FF>;
    constexpr int TS = W::TS, NN = W::NN, THREADS = W::THREADS;
    constexpr int STEM = POST ? 9 : 25, STEM_STEPS = (STEM + 3) / 4;
    _Float16 *hiP = reinterpret_cast<_Float16 *>(lds), *loP = hiP + W::LO;
    float *brd = lds + W::PLANES_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pn = lane & 15;              // result layout: channels 4 g + r of the group at tile position pn
    const int cgroup = wave % NN, mq = wave / NN;
    const int ch4 = cgroup * 16 + 4 * g;                  // the lane's first channel

    for (int i = tid; i < W::LDS_FLOATS / 4; i += THREADS) reinterpret_cast<f32x4 *>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int i = tid; i < 400; i += THREADS) brd[(i / 10 + 2) * BW + (i % 10) + 2] = boards[board * 400 + i];
    __syncthreads();

    int base[TS];                   // cell offset (in halves) of the top-left position of the 3x3 window of the lane's position
#pragma unroll
    for (int s = 0; s < TS; s++) {
        int T = mq + 2 * s; if (T >= NT) T = NT - 1;
        const int p = 16 * T + pn;
        base[s] = ((p / 10) * SPW + (p % 10)) * 8;
    }

    // stem (one input channel): im2col GEMM on the fp32 matrix instruction as in k_trunk16s -- A = w[channel lane & 15 of the
    // group][tap 4 st + g], B = the board value under that tap at the lane's position
    f32x4 x[TS];
    {
        float wA[STEM_STEPS];
        int toff[STEM_STEPS];
#pragma unroll
        for (int st = 0; st < STEM_STEPS; st++) {
            const int t = 4 * st + g, tc = t < STEM ? t : STEM - 1;
            wA[st] = t < STEM ? wts[(cgroup * 16 + pn) * STEM + tc] : 0.f;
            toff[st] = POST ? (tc / 3 + 1) * BW + (tc % 3) + 1 : (tc / 5) * BW + (tc % 5);
        }
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f};
        if (POST) b0 = *reinterpret_cast<const f32x4 *>(wts + FF * STEM + ch4);
#pragma unroll
        for (int s = 0; s < TS; s++) {
            int T = mq + 2 * s; if (T >= NT) T = NT - 1;
            const int p = 16 * T + pn, pb = (p / 10) * BW + (p % 10);
            f32x4 c = b0;
#pragma unroll
            for (int st = 0; st < STEM_STEPS; st++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[st], brd[pb + toff[st]], c, 0, 0, 0);
            x[s] = POST ? relu4(c) : c;
        }
    }

    // the lane's four channels live in oct-plane ch4 / 8 at sub-channel ch4 % 8 (= 0 or 4); its fragments: plane 4 kg + g
    _Float16 *ownh = hiP + (2 * cgroup + (g >> 1)) * W::PLANEH + (SPW + 1) * 8 + 4 * (g & 1), *ownl = ownh + W::LO;
    const _Float16 *hiL = hiP + g * W::PLANEH, *loL = loP + g * W::PLANEH;
    const float *wb = wts + FF * STEM + (POST ? FF : 0);
    f32x4 am[TS];
    float act_max = 0.f;            // largest activation this lane has split (store_split4)
#ifdef TRL_PROFILE
    unsigned long long tprof[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TPROF_T(tp_begin);
    for (int blk = 0; blk < n_blocks; blk++)
#pragma unroll
    for (int half = 0; half < 2; half++) {            // the two convs of a block are separate code: nothing is selected at run time
        TPROF_T(tp0);
        const bool first = half == 0;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f}, bias = sh;
        const float *wq;
        if (POST) {                                    // block: b1[F] W1 b2[F] W2
            bias = *reinterpret_cast<const f32x4 *>(first ? wb + ch4 : wb + FF + W::CONVW + ch4);
            wq = wb + (first ? FF : 2 * FF + W::CONVW);
        } else {                                       // block: s1[F] t1[F] Wa ba[F] Wb
#if TRL_EXP != 4
            if (first) {
                sc = *reinterpret_cast<const f32x4 *>(wb + ch4); sh = *reinterpret_cast<const f32x4 *>(wb + FF + ch4);
                bias = *reinterpret_cast<const f32x4 *>(wb + 2 * FF + W::CONVW + ch4);
            }
#endif
            wq = wb + (first ? 2 * FF : 3 * FF + W::CONVW);
        }
        const uint4 *wq4 = reinterpret_cast<const uint4 *>(wq) + cgroup * 64 + lane;
        const f16x8 wh = __builtin_bit_cast(f16x8, wq4[0]), wl = __builtin_bit_cast(f16x8, wq4[NN * 64]);
        // conv input -> planes (own positions), split: the block input (pre-activation: through bn1 + relu) for the first
        // conv, relu(previous conv) for the second
#pragma unroll
        for (int s = 0; s < (TRL_EXP == 3 ? 1 : TS); s++) {
            if (s < 12 || mq == 0) {                   // (tile 25 does not exist)
                const f32x4 v = relu4(first ? (POST ? x[s] : __builtin_elementwise_fma(x[s], sc, sh)) : am[s]);
                store_split4(ownh, ownl, base[s], v, act_max);
            }
        }
        TPROF_T(tp1);
        __syncthreads();
        TPROF_T(tp2);
#pragma unroll
        for (int s = 0; s < TS; s++) am[s] = bias * SPLIT;
        if (mq == 0) conv_wide_s<FF, 13>(hiL, loL, wq4, wh, wl, base, am);
        else conv_wide_s<FF, 12>(hiL, loL, wq4, wh, wl, base, am);
#pragma unroll
        for (int s = 0; s < TS; s++) am[s] *= INV_SPLIT;
        TPROF_T(tp3);
        if (!first) {
#pragma unroll
            for (int s = 0; s < TS; s++) {
                x[s] += am[s];
                if (POST) x[s] = relu4(x[s]);
            }
            wb += POST ? 2 * FF + 2 * W::CONVW : 3 * FF + 2 * W::CONVW;
        }
        TPROF_T(tp4);
        __syncthreads();                              // every wave has read its inputs: the planes can be overwritten
        TPROF_T(tp5);
        TPROF_ADD(0, tp0, tp1); TPROF_ADD(1, tp1, tp2); TPROF_ADD(2, tp2, tp3); TPROF_ADD(3, tp3, tp4); TPROF_ADD(4, tp4, tp5);
    }
#ifdef TRL_PROFILE
    {
        TPROF_T(tp_end);
        tprof[8] = tp_end - tp_begin;
        if ((blockIdx.x == 0 || blockIdx.x == 300) && lane == 0 && wave < 8)
            for (int i = 0; i < 10; i++) g_tprof[((blockIdx.x ? 1 : 0) * 8 + wave) * 10 + i] = tprof[i];
    }
#endif

    report_split_range(act_max);
    float *part = lds;                                 // [kout][NN][400] partial sums (the planes are dead now)
    if (POST) {
        for (int k = 0; k < kout; k++) {               // tail: kout 1x1 projections of the feature map, raw
            const f32x4 wk = *reinterpret_cast<const f32x4 *>(wb + k * FF + ch4);
#pragma unroll
            for (int sidx = 0; sidx < TS; sidx++) {
                const int T = mq + 2 * sidx;
                float v = x[sidx][0] * wk[0];
                v = fmaf(x[sidx][1], wk[1], v); v = fmaf(x[sidx][2], wk[2], v); v = fmaf(x[sidx][3], wk[3], v);
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (g == 0 && T < NT) part[(k * NN + cgroup) * 400 + 16 * T + pn] = v;
            }
        }
        __syncthreads();
        for (int i = tid; i < 400 * kout; i += THREADS) {
            const int k = i / 400, pos = i % 400;
            float v = 0.f;
#pragma unroll
            for (int n = 0; n < NN; n++) v += part[(k * NN + n) * 400 + pos];
            out[(board * kout + k) * 400 + pos] = v;
        }
    } else {
        // tail: relu(bn(x)) -> 1x1 conv over the FF channels (4 in the lane, 4 lane groups, NN waves) -> folded bn -> relu
        const f32x4 s = *reinterpret_cast<const f32x4 *>(wb + ch4), t = *reinterpret_cast<const f32x4 *>(wb + FF + ch4);
        const f32x4 wk = *reinterpret_cast<const f32x4 *>(wb + 2 * FF + ch4);
        const float bk = wb[3 * FF];
#pragma unroll
        for (int sidx = 0; sidx < TS; sidx++) {
            const int T = mq + 2 * sidx;
            const f32x4 y = relu4(x[sidx] * s + t);
            float v = y[0] * wk[0];
            v = fmaf(y[1], wk[1], v); v = fmaf(y[2], wk[2], v); v = fmaf(y[3], wk[3], v);
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (g == 0 && T < NT) part[cgroup * 400 + 16 * T + pn] = v;
        }
        __syncthreads();
        for (int i = tid; i < 400; i += THREADS) {
            float v = bk;
#pragma unroll
            for (int n = 0; n < NN; n++) v += part[n * 400 + i];
            out[board * 400 + i] = fmaxf(v, 0.f);
        }
    }
}

// COMPACT = false: workgroup b takes board b (boards with need == 0 end at once).
// COMPACT = true (launches with need flags): workgroup b takes the b-th LIVE board.  A workgroup of this kernel owns its CU
// (138 KB of LDS at 64 filters), so a workgroup that only looks at its flag still queues for a CU behind the ones that
// work; with the live boards scattered over the grid a search tick whose boards are mostly cached (the search's board table:
// 60-70 % at MAX_ITER = 800) pays for rounds of workgroups it does not need.  With the live boards in front the dispatcher
// packs them ceil(live / CUs) deep and the idle workgroups run out behind them.  Every workgroup builds the list for itself
// from the need flags (n_boards bytes, one read, two barriers).
template <int FF, bool POST, bool COMPACT>
__global__ __launch_bounds__((WideS<FF>::THREADS), 2) void k_trunk_wide_s(const float *__restrict__ boards,
                                                                       const float *__restrict__ wts, int n_blocks,
                                                                       float *__restrict__ out,
                                                                       const uint8_t *__restrict__ need, int kout, int n_boards) {
    using W = WideS<FF>;
    __shared__ __attribute__((aligned(16))) float lds[W::LDS_FLOATS];
    if (!COMPACT) {
        if (need && !need[blockIdx.x]) return;
        trunk_wide_s_board<FF, POST>(boards, wts, n_blocks, out, kout, (size_t)blockIdx.x, lds);
        return;
    }
    __shared__ int wave_cnt[W::THREADS / 64], mine_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) mine_sh = -1;
    int seen = 0;                                                          // live boards before this chunk
    for (int base = 0; base < n_boards && seen <= (int)blockIdx.x; base += W::THREADS) {
        const int b = base + tid;
        const bool on = b < n_boards && need[b] != 0;
        const uint64_t m = __ballot(on);
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = seen, total = seen;
        for (int w = 0; w < W::THREADS / 64; w++) { const int c = wave_cnt[w]; if (w < wave) off += c; total += c; }
        if (on && off + __popcll(m & ((1ull << lane) - 1ull)) == (int)blockIdx.x) mine_sh = b;      // the blockIdx-th live board
        __syncthreads();
        seen = total;
    }
    const int board = mine_sh;
    if (board < 0) return;                                                 // fewer live boards than workgroups: nothing to do
    __syncthreads();
    trunk_wide_s_board<FF, POST>(boards, wts, n_blocks, out, kout, (size_t)board, lds);
}

}  // namespace

extern "C" {

// process_grid of AlphaSame (architectures.py:121-127) for filters = 16, kernels = 1.
// boards: device float[n_boards][400]; packed: device float[400 + n_blocks * 4656 + 49] in the layout
// nets.AlphaSame.prepare() builds; out: device float[n_boards][400]; stream: hipStream_t.
int trl_alphasame_trunk16(const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                          float *dev_out, void *stream) {
    return trl_alphasame_trunk16_need(dev_boards, n_boards, dev_packed, n_blocks, dev_out, nullptr, stream);
}

int trl_alphasame_trunk16_need(const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                               float *dev_out, const uint8_t *dev_need, void *stream) {
    if (!dev_boards || !dev_packed || !dev_out || n_boards <= 0 || n_blocks < 0)
        return fail(TRL_E_INVALID, "trl_alphasame_trunk16: %s", "bad argument");
    hipLaunchKernelGGL(k_trunk16, dim3(n_boards), dim3(TRUNK_THREADS), 0, (hipStream_t)stream, dev_boards, dev_packed,
                       n_blocks, dev_out, dev_need);
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

// The 16-filter tower on the f16 matrix rate with every fp32 operand split into two halves (k_trunk16s above): same
// contract as trl_alphasame_trunk16_need; dev_packed in the layout of nets.AlphaSame._pack_tower_split().
int trl_alphasame_trunk16_split(const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                                float *dev_out, const uint8_t *dev_need, void *stream) {
    if (!dev_boards || !dev_packed || !dev_out || n_boards <= 0 || n_blocks < 0)
        return fail(TRL_E_INVALID, "trl_alphasame_trunk16_split: %s", "bad argument");
    // up to one board per CU the fragment reads are interleaved with the matrix instructions, beyond that grouped (see
    // conv_step); TRL_SPLIT_ILV = 0 / 1 forces one form (probe knob).  Three boards per CU (168 registers, measured with
    // the same kernel) ran at the speed of two: 768 boards 89 against 96 us, 1536 169 against 172.
    static const int force = getenv("TRL_SPLIT_ILV") ? atoi(getenv("TRL_SPLIT_ILV")) : -1;
    // (with need flags about half of the boards are skipped at once: the tick of a 512-slot group has ~512 live boards of
    // 1024, and there the interleaved form was 1.5 % faster as well, beside the other groups' kernels)
    if (force < 0 ? n_boards <= (dev_need ? 1024 : 256) : force != 0)
        hipLaunchKernelGGL((k_trunk16s<2, true>), dim3(n_boards), dim3(TRUNK_THREADS), 0, (hipStream_t)stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need);
    else
        hipLaunchKernelGGL((k_trunk16s<2, false>), dim3(n_boards), dim3(TRUNK_THREADS), 0, (hipStream_t)stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need);
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

// The split 16-filter tower of one search tick together with the search's legal-move job (trl_mcts_legal_in_tower): the
// boards are the grids trl_mcts_select_need / trl_mcts_expand_alphasame_select wrote, dev_need their flags; runs on the
// engine's stream.
int trl_mcts_tower16_legal(trl_engine *h, const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                           float *dev_out, const uint8_t *dev_need) {
    if (!h || !dev_boards || !dev_packed || !dev_out || !dev_need || n_boards <= 0 || n_blocks < 0)
        return fail(TRL_E_INVALID, "trl_mcts_tower16_legal: %s", "bad argument");
    LeafLegalJob job;
    if (!trl_mcts_leaf_legal_job(h, &job)) return fail(TRL_E_STATE, "trl_mcts_tower16_legal: %s", "trl_mcts_legal_in_tower is not on");
    HIPCHK(hipSetDevice(h->cfg.device));
    static const int force = getenv("TRL_SPLIT_ILV") ? atoi(getenv("TRL_SPLIT_ILV")) : -1;
    if (force < 0 ? n_boards <= 1024 : force != 0)
        hipLaunchKernelGGL((k_trunk16s<2, true, true>), dim3(job.n_slots + n_boards), dim3(TRUNK_THREADS), 0, h->stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need, job);
    else
        hipLaunchKernelGGL((k_trunk16s<2, false, true>), dim3(job.n_slots + n_boards), dim3(TRUNK_THREADS), 0, h->stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need, job);
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

// process_grid of AlphaSame for filters in {16, 32, 64}, kernels = 1: the same contract as trl_alphasame_trunk16 with
// the packed weights in the general layout of nets.AlphaSame._pack() (identical to the 16-filter layout at filters = 16).
int trl_alphasame_trunk(const float *dev_boards, int n_boards, const float *dev_packed, int filters, int n_blocks,
                        float *dev_out, const uint8_t *dev_need, void *stream) {
    if (!dev_boards || !dev_packed || !dev_out || n_boards <= 0 || n_blocks < 0)
        return fail(TRL_E_INVALID, "trl_alphasame_trunk: %s", "bad argument");
    if (filters == 16) return trl_alphasame_trunk16_need(dev_boards, n_boards, dev_packed, n_blocks, dev_out, dev_need, stream);
    if (filters == 32)
        hipLaunchKernelGGL((k_trunk_wide<32, false>), dim3(n_boards), dim3(Wide<32>::THREADS), 0, (hipStream_t)stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need, 1);
    else if (filters == 64)
        hipLaunchKernelGGL((k_trunk_wide<64, false>), dim3(n_boards), dim3(Wide<64>::THREADS), 0, (hipStream_t)stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need, 1);
    else
        return fail(TRL_E_INVALID, "trl_alphasame_trunk: %s", "the fused tower exists for 16, 32 and 64 filters");
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

// The wide towers at the f16 matrix rate with split operands (k_trunk_wide_s): same contracts as trl_alphasame_trunk /
// trl_resnet_trunk for filters 32 and 64, with the 3x3 convolutions packed by nets._pack_conv3x3_split_wide.
// resnet != 0: the BaseResNet family (kout raw projections), else AlphaSame (kout ignored).
int trl_trunk_wide_split(const float *dev_boards, int n_boards, const float *dev_packed, int filters, int n_blocks, int resnet,
                         int kout, float *dev_out, const uint8_t *dev_need, void *stream) {
    if (!dev_boards || !dev_packed || !dev_out || n_boards <= 0 || n_blocks < 0 || (resnet && (kout < 1 || kout > 8)))
        return fail(TRL_E_INVALID, "trl_trunk_wide_split: %s", "bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (filters != 32 && filters != 64)
        return fail(TRL_E_INVALID, "trl_trunk_wide_split: %s", "the split wide tower exists for 32 and 64 filters");
    // with need flags: the live boards first (k_trunk_wide_s<.., COMPACT>); TRL_WIDE_COMPACT=0: workgroup b = board b
    static const int compact_on = getenv("TRL_WIDE_COMPACT") ? atoi(getenv("TRL_WIDE_COMPACT")) : 1;
    const bool persist = compact_on && dev_need;
    const int grid = n_boards;
#define TRL_LAUNCH_WS(FF_, POST_) do { if (persist) hipLaunchKernelGGL((k_trunk_wide_s<FF_, POST_, true>), dim3(grid), dim3(WideS<FF_>::THREADS), \
                                                     0, st, dev_boards, dev_packed, n_blocks, dev_out, dev_need, resnet ? kout : 1, n_boards); \
                                       else hipLaunchKernelGGL((k_trunk_wide_s<FF_, POST_, false>), dim3(grid), dim3(WideS<FF_>::THREADS), \
                                                     0, st, dev_boards, dev_packed, n_blocks, dev_out, dev_need, resnet ? kout : 1, n_boards); } while (0)
    if (filters == 64) { if (resnet) TRL_LAUNCH_WS(64, true); else TRL_LAUNCH_WS(64, false); }
    else { if (resnet) TRL_LAUNCH_WS(32, true); else TRL_LAUNCH_WS(32, false); }
#undef TRL_LAUNCH_WS
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

// Tower of the BaseResNet family (architectures.py:173-353: stem conv3x3 + BN + ReLU, n_blocks post-activation residual
// blocks) for filters in {32, 64}, followed by kout (1..8) raw 1x1 projections of the final feature map:
// dev_out float[n_boards][kout][400].  dev_packed (batch norms folded into the convolutions): stem W[F][9] b[F];
// per block b1[F] W1 b2[F] W2 (3x3 convs packed as for trl_alphasame_trunk); tail Wk[kout][F]; >= 16 KiB padding.
int trl_resnet_trunk(const float *dev_boards, int n_boards, const float *dev_packed, int filters, int n_blocks, int kout,
                     float *dev_out, const uint8_t *dev_need, void *stream) {
    if (!dev_boards || !dev_packed || !dev_out || n_boards <= 0 || n_blocks < 0 || kout < 1 || kout > 8)
        return fail(TRL_E_INVALID, "trl_resnet_trunk: %s", "bad argument");
    if (filters == 32)
        hipLaunchKernelGGL((k_trunk_wide<32, true>), dim3(n_boards), dim3(Wide<32>::THREADS), 0, (hipStream_t)stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need, kout);
    else if (filters == 64)
        hipLaunchKernelGGL((k_trunk_wide<64, true>), dim3(n_boards), dim3(Wide<64>::THREADS), 0, (hipStream_t)stream, dev_boards,
                           dev_packed, n_blocks, dev_out, dev_need, kout);
    else
        return fail(TRL_E_INVALID, "trl_resnet_trunk: %s", "the fused tower exists for 32 and 64 filters");
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

// Range guard of the split towers (k_trunk16s / k_trunk_wide_s): the number of boards, since the last reset, in which an
// activation reached the largest finite f16 (65 504) when it was split into halves -- their outputs are not to be trusted
// (inf / NaN inside the tower); the fp32 towers (trl_alphasame_trunk / trl_resnet_trunk) have no such limit.  The caller
// synchronises the stream(s) its towers ran on first; `device` = HIP device ordinal.
int trl_tower_overflow(int device, int reset, uint32_t *count_out) {
    if (!count_out) return fail(TRL_E_INVALID, "trl_tower_overflow: %s", "null argument");
    HIPCHK(hipSetDevice(device));
    unsigned int c = 0;
    HIPCHK(hipMemcpyFromSymbol(&c, HIP_SYMBOL(g_split_overflow), sizeof c));
    if (reset && c) { const unsigned int z = 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_split_overflow), &z, sizeof z)); }
    *count_out = c;
    return TRL_OK;
}

#ifdef TRL_PROFILE
int trl_debug_tower_prof(unsigned long long *host_out) {
    HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tprof), sizeof(unsigned long long) * 160));
    return TRL_OK;
}
#endif

}  // extern "C"
