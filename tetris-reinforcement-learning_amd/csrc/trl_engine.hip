// trl_engine.hip -- kernels + the C ABI of include/tetris_engine.h (gfx950 only, no CPU path).
//
// Kernel geometry: one 128-thread workgroup (two wavefronts) per game.  Wavefront w searches
// the placements of piece slot w (active piece / hold-or-next piece); thread 0 then applies
// the chosen placement.  The game's 320-byte state lives in LDS for the whole launch (the
// rollout kernel keeps it there across plies), global traffic is one coalesced 320-byte read
// and write per game per launch plus the move list when it is asked for.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include "trl_internal.h"

using namespace trl;

static_assert(sizeof(trl_player_state) == 144, "trl_player_state layout");
static_assert(sizeof(trl_game_state) == 320, "trl_game_state layout");

namespace {

// ---- kernels ------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_reset(trl_game_state *states, const uint32_t *streams, int ruleset, uint64_t seed, int n) {
    int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    trl_game_state st;
    game_setup(st, ruleset, seed, streams ? streams[g] : (uint32_t)g);
    states[g] = st;
}

__global__ __launch_bounds__(BLOCK, 8) void k_legal(trl_game_state *states, uint16_t *counts, uint16_t *moves) {
    __shared__ BlockShared sh;
    const int tid = threadIdx.x, game = blockIdx.x;
    load_state(sh, states + game, tid);
    __syncthreads();
    int total = block_movegen(sh, tid);
    for (int i = tid; i < total; i += BLOCK) moves[(size_t)game * TRL_MAX_LEGAL + i] = merged_move(sh, i);
    if (tid == 0) {
        counts[game] = (uint16_t)total;
        if (sh.overflow) states[game].flags |= TRL_FLAG_OVERFLOW;
    }
}

// get_move_matrix(player, 'brute-force') -- the reference's default argument (move_generation.py:752); row M6.
__global__ __launch_bounds__(BLOCK) void k_legal_bruteforce(trl_game_state *states, uint16_t *counts, uint16_t *moves) {
    __shared__ BfBlockShared sh;
    const int tid = threadIdx.x, game = blockIdx.x;
    {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(states + game);
        uint32_t *d = reinterpret_cast<uint32_t *>(&sh.g);
        if (tid < STATE_DW) d[tid] = s[tid];
    }
    __syncthreads();
    const int total = block_movegen(sh, tid);
    for (int i = tid; i < total; i += BLOCK) moves[(size_t)game * TRL_MAX_LEGAL + i] = merged_move(sh, i);
    if (tid == 0) {
        counts[game] = (uint16_t)total;
        if (sh.overflow) states[game].flags |= TRL_FLAG_OVERFLOW;
    }
}

__global__ __launch_bounds__(BLOCK) void k_step(trl_game_state *states, const uint16_t *in_moves, uint64_t seed, int add_bag, int check_legal) {
    __shared__ BlockShared sh;
    const int tid = threadIdx.x, game = blockIdx.x;
    const int move = in_moves[game];
    if (move == TRL_MOVE_NONE) return;
    load_state(sh, states + game, tid);
    __syncthreads();
    bool legal = true;
    if (check_legal) {
        int total = block_movegen(sh, tid);
        bool mine = false;
        for (int i = tid; i < total; i += BLOCK) mine |= (merged_move(sh, i) == move);
        // (an explicit flag, not __syncthreads_or: the device library's workgroup reduction returned 0 for a set predicate in
        // a few of 8 092 workgroups per launch once the two waves' searches finished far apart)
        if (tid == 0) sh.done = 0;
        __syncthreads();
        if (mine) sh.done = 1;
        __syncthreads();
        legal = sh.done != 0;
    }
    if (tid < 64) {
        if (legal) make_move(sh.g, seed, move, add_bag != 0, tid);
        else if (tid == 0) sh.g.flags |= TRL_FLAG_BADMOVE;
        if (tid == 0 && check_legal && sh.overflow) sh.g.flags |= TRL_FLAG_OVERFLOW;
    }
    __syncthreads();
    store_state(sh, states + game, tid);
}

#ifdef TRL_SLOT_TIMES
__device__ unsigned long long g_slot_t[2 * 16384];     // per slot: wall clock at the start / end of its part of the last rollout launch
#endif

// C2 workload: n_plies x { legal -> uniform pick -> make_move(add_bag) -> re-setup when finished }
// Pacing.  Every game plays its plies in sequence and all slots are resident at once (16 workgroups per CU).  The issue
// arbiter of a SIMD serves its OLDEST ready wave first, so the workgroups dispatched first run up to twice as fast as
// the last ones: unpaced, the first slot of a 1000-ply launch is done after 21.7 ms, the median after 30 ms, the last
// after 42.9 ms, the chip emptying out all the while (profiles/r02/r02c_rollout_tail.txt).  Slots therefore publish
// their progress and set their waves' issue priority (s_setprio) by how far behind the others they are.  Scheduling
// only -- the games are the same bit for bit.
//   After every ply a slot stores its ply count in prog[] and reads the 64 counts of its comparison set (one per lane,
//   from L2, no shared counter); its rank in the set (members that are further behind) picks the priority: the r3
//   least advanced get 3, the next up to r2 get 2, up to r1 get 1, the rest 0 (`rankt` = r3 | r2 << 8 | r1 << 16;
//   0 = pacing off).  The set of slot g is {g % S + j * S, j = 0..63}, S = ceil(slots / 64): 64 slots spread evenly over
//   the dispatch order -- neighbours in the grid have the same age, and comparing a slot with them shows nothing.
//   Round 1 used one global counter (an atomic per slot and 16 plies, ~30 ns each, serialised): as good on launches of
//   1000 plies (34.0 us per ply against 35.1 here) but too slow to react on short ones (50 us per ply at 20 plies per
//   launch against 40.7 here; profiles/r02/r02a_rollout_regime.txt, r02f_rollout_tail.txt).  What remains at 20 plies
//   is the critical path of the most expensive games: even with the SIMD to itself a wave needs ~25 us for a ply.
__global__ __launch_bounds__(BLOCK, 8) void k_rollout(trl_game_state *states, uint64_t seed, int n_plies, int max_plies,
                                                   unsigned long long *steps_total, unsigned int *prog, int rankt,
                                                   unsigned long long *steps_next, unsigned int *prog_next) {
    __shared__ BlockShared sh;
    const int tid = threadIdx.x, game = blockIdx.x;
    if (tid == 0) {                 // the next launch's counters (the other block: nobody reads it during this launch)
        const int S = ((int)gridDim.x + 63) >> 6;
        prog_next[(game % S) * 64 + game / S] = 0;
        if (game == 0) *steps_next = 0;
    }
#ifdef TRL_SLOT_TIMES
    const unsigned long long t_begin = wall_clock64();
#endif
    load_state(sh, states + game, tid);
    __syncthreads();
    int steps = 0;
    for (int k = 0; k < n_plies; k++) {
        if (rankt && k) {
            // both waves look for themselves (no barrier); the counts of a set are contiguous in prog[] (256 bytes)
            const int lane = tid & 63, S = ((int)gridDim.x + 63) >> 6, set = game % S;
            const bool member = set + lane * S < (int)gridDim.x;
            const unsigned int theirs = member ? __hip_atomic_load(prog + set * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : (unsigned int)k;
            const int rank = (int)__popcll(__ballot(member && theirs < (unsigned int)k));   // members that are behind this slot
            if (rank < (rankt & 255)) __builtin_amdgcn_s_setprio(3);
            else if (rank < ((rankt >> 8) & 255)) __builtin_amdgcn_s_setprio(2);
            else if (rank < ((rankt >> 16) & 255)) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        // the thread index through an opaque move, once per ply: everything the search derives from it (lane, rotation, map
        // column, LDS addresses, a dozen lane predicates) is then recomputed per ply instead of being hoisted out of the ply
        // loop and kept in registers for the whole launch -- at 64 registers per lane that hoisting cost 9 spilled registers
        // (19 MB of scratch written per launch, more than the game states themselves)
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));
        int total = block_movegen(sh, tid_p);
        if (tid < 64) {
            trl_game_state &g = sh.g;
            if (tid == 0 && sh.overflow) g.flags |= TRL_FLAG_OVERFLOW;
            if (total > 0) {
                int idx = (int)rng_below(seed, g.stream, TRL_RNG_POLICY, g.pol_ctr, (uint32_t)total);
                wave_sync();
                if (tid == 0) g.pol_ctr++;
                make_move(g, seed, merged_move(sh, idx), true, tid_p);
                steps++;
            }
            if (tid == 0 && (total == 0 || is_terminal(g) || (int)g.ply >= max_plies)) {
                uint32_t stream = g.stream + 0x01000000u, episode = g.episode + 1;
                int ruleset = g.ruleset; uint8_t flags = g.flags;
                // episode e runs on stream + e * 2^24 in uint32: the 256th re-setup lands on the slot's first stream again and
                // the slot replays its games from there -- said in the state (the search refuses this up front, trl_mcts_create)
                if (episode >= 256u) flags |= TRL_FLAG_STREAM_WRAP;
                game_setup(g, ruleset, seed, stream);
                g.episode = episode; g.flags = flags;
            }
            if (rankt && tid == 0) {
                const int S = ((int)gridDim.x + 63) >> 6;
                __hip_atomic_store(prog + (game % S) * 64 + game / S, (unsigned int)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    store_state(sh, states + game, tid);
    if (tid == 0 && steps_total) atomicAdd(steps_total, (unsigned long long)steps);
#ifdef TRL_SLOT_TIMES
    if (tid == 0) { g_slot_t[2 * game] = t_begin; g_slot_t[2 * game + 1] = wall_clock64(); }    // 100 MHz chip-wide clock
#endif
}

// exhaustive check of the integer attack formulas against stats.py (tests only)
__global__ void k_attack_table(const int16_t *rows, int n, int16_t *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int16_t *r = rows + 8 * i;
    trl_player_state p;
    memset(&p, 0, sizeof p);
    p.combo = r[5]; p.b2b = r[6]; p.b2b_level = (uint8_t)r[7];
    int a = get_attack(p, r[0], r[1], r[2] != 0, r[3] != 0, r[4] != 0);
    out[4 * i + 0] = (int16_t)a; out[4 * i + 1] = p.combo; out[4 * i + 2] = p.b2b; out[4 * i + 3] = p.b2b_level;
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------
thread_local char trl_g_err[512] = "";

extern "C" {

const char *trl_last_error(void) { return trl_g_err; }
const char *trl_version(void) { return "trl-hip 0.1 (gfx950)"; }

int trl_engine_create(const trl_config *cfg, trl_engine **out) {
    if (!cfg || !out || cfg->n_games <= 0 || (cfg->ruleset != TRL_RULESET_S1 && cfg->ruleset != TRL_RULESET_S2))
        return fail(TRL_E_INVALID, "trl_engine_create: %s", "bad config");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg->device >= ndev)
        return fail(TRL_E_NO_DEVICE, "trl_engine_create: %s", "no HIP device visible (the engine has no CPU path)");
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(TRL_E_NO_DEVICE, "trl_engine_create: device is %s, this library is built for gfx950 only", prop.gcnArchName);
    trl_engine *h = new (std::nothrow) trl_engine();
    if (!h) return fail(TRL_E_INVALID, "trl_engine_create: %s", "out of host memory");
    memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    h->n_cu = prop.multiProcessorCount;
    auto alloc = [&]() -> int {
        if (cfg->stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
        else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
        size_t n = (size_t)cfg->n_games;
        HIPCHK(hipMalloc(&h->d_states, n * sizeof(trl_game_state)));
        HIPCHK(hipMalloc(&h->d_counts, n * sizeof(uint16_t)));
        HIPCHK(hipMalloc(&h->d_moves, n * TRL_MAX_LEGAL * sizeof(uint16_t)));
        HIPCHK(hipMalloc(&h->d_in_moves, n * sizeof(uint16_t)));
        // two blocks of [env-steps of the launch (+ 8 bytes of padding)][plies played in the launch per slot (rollout
        // pacing)]: a rollout launch counts in one and zeroes the other for its successor -- no memset between launches
        h->pp_stride = 16 + (n + 63) / 64 * 64 * sizeof(unsigned int);
        HIPCHK(hipMalloc(&h->d_steps, 2 * h->pp_stride));
        HIPCHK(hipMemset(h->d_steps, 0, 2 * h->pp_stride));
        h->d_prog = reinterpret_cast<unsigned int *>(reinterpret_cast<char *>(h->d_steps) + 16);
        h->pp = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&h->h_steps), 64));
        HIPCHK(hipEventCreate(&h->ev0));
        HIPCHK(hipEventCreate(&h->ev1));
        return TRL_OK;
    };
    int rc = alloc();
    if (rc != TRL_OK) {                            // release what was allocated before the failure
        hipFree(h->d_states); hipFree(h->d_counts); hipFree(h->d_moves); hipFree(h->d_in_moves); hipFree(h->d_steps);
        if (h->h_steps) hipHostFree(h->h_steps);
        if (h->ev0) hipEventDestroy(h->ev0);
        if (h->ev1) hipEventDestroy(h->ev1);
        if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
        delete h;
        return rc;
    }
    *out = h;
    return TRL_OK;
}

void trl_engine_destroy(trl_engine *h) {
    if (!h) return;
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    trl_mcts_free_internal(h);
    if (h->graph_many_exec) hipGraphExecDestroy(h->graph_many_exec);
    if (h->graph_many) hipGraphDestroy(h->graph_many);
    if (h->graph_exec) hipGraphExecDestroy(h->graph_exec);
    if (h->graph) hipGraphDestroy(h->graph);
    hipFree(h->d_states); hipFree(h->d_counts); hipFree(h->d_moves); hipFree(h->d_in_moves); hipFree(h->d_steps);
    if (h->h_steps) hipHostFree(h->h_steps);
    hipEventDestroy(h->ev0); hipEventDestroy(h->ev1);
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
}

int trl_set_stream(trl_engine *h, void *hip_stream) {
    if (!h) return fail(TRL_E_INVALID, "%s", "null engine");
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream) { HIPCHK(hipStreamDestroy(h->stream)); h->own_stream = false; }
    if (hip_stream) h->stream = (hipStream_t)hip_stream;
    else { HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    return TRL_OK;
}

// ---- HIP graph of the calls issued on the engine's stream between _begin and _end ---------------------------------
// (the search tick is three or four launch-bound kernels: replaying it as one graph removes the per-launch host cost).
// Two graphs: the calls of ONE tick (trl_graph_end) and, optionally, the same calls issued `ticks` times in a row
// (trl_graph_end_many): between two graph launches the stream idles for ~8 us (kernel trace of the C3 tick: expand ends,
// the next select starts 8.7 us later; 4.3 us between child-graph nodes), inside a flat graph the next kernel follows at
// once.  trl_graph_launch(n) replays the long graph n / ticks times and the single one for the rest.
int trl_graph_begin(trl_engine *h) {
    if (!h) return fail(TRL_E_INVALID, "%s", "null engine");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    return TRL_OK;
}

int trl_graph_end(trl_engine *h) {
    if (!h) return fail(TRL_E_INVALID, "%s", "null engine");
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamEndCapture(h->stream, &g));
    if (h->graph_many_exec) { HIPCHK(hipGraphExecDestroy(h->graph_many_exec)); h->graph_many_exec = nullptr; }   // of an older tick
    if (h->graph_many) { HIPCHK(hipGraphDestroy(h->graph_many)); h->graph_many = nullptr; }
    if (h->graph_exec) { HIPCHK(hipGraphExecDestroy(h->graph_exec)); h->graph_exec = nullptr; }
    if (h->graph) { HIPCHK(hipGraphDestroy(h->graph)); h->graph = nullptr; }
    h->graph = g;
    h->graph_unroll = 1;
    HIPCHK(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    return TRL_OK;
}

int trl_graph_end_many(trl_engine *h, int ticks) {
    if (!h || ticks < 2) return fail(TRL_E_INVALID, "%s", "trl_graph_end_many: null engine / fewer than 2 ticks");
    hipGraph_t g = nullptr;
    HIPCHK(hipStreamEndCapture(h->stream, &g));
    if (!h->graph_exec) { hipGraphDestroy(g); return fail(TRL_E_STATE, "%s", "trl_graph_end_many before trl_graph_end"); }
    if (h->graph_many_exec) { HIPCHK(hipGraphExecDestroy(h->graph_many_exec)); h->graph_many_exec = nullptr; }
    if (h->graph_many) { HIPCHK(hipGraphDestroy(h->graph_many)); h->graph_many = nullptr; }
    h->graph_many = g;
    HIPCHK(hipGraphInstantiate(&h->graph_many_exec, h->graph_many, nullptr, nullptr, 0));
    h->graph_unroll = ticks;
    return TRL_OK;
}

int trl_graph_launch(trl_engine *h, int n_times) {
    if (!h || !h->graph_exec) return fail(TRL_E_STATE, "%s", "trl_graph_launch without a captured graph");
    HIPCHK(hipSetDevice(h->cfg.device));
    int left = n_times;
    if (h->graph_many_exec)
        for (; left >= h->graph_unroll; left -= h->graph_unroll) HIPCHK(hipGraphLaunch(h->graph_many_exec, h->stream));
    for (; left > 0; left--) HIPCHK(hipGraphLaunch(h->graph_exec, h->stream));
    return TRL_OK;
}

int trl_env_reset(trl_engine *h, const uint32_t *stream_ids) {
    if (!h) return fail(TRL_E_INVALID, "%s", "null engine");
    HIPCHK(hipSetDevice(h->cfg.device));
    int n = h->cfg.n_games;
    uint32_t *d_streams = nullptr;
    if (stream_ids) {
        HIPCHK(hipMalloc(&d_streams, (size_t)n * 4));
        hipError_t ec = hipMemcpyAsync(d_streams, stream_ids, (size_t)n * 4, hipMemcpyHostToDevice, h->stream);
        if (ec != hipSuccess) { (void)hipFree(d_streams); (void)hipGetLastError(); return fail(TRL_E_HIP, "trl_env_reset: %s", hipGetErrorString(ec)); }
    }
    hipLaunchKernelGGL(k_reset, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, h->stream,
                       h->d_states, d_streams, h->cfg.ruleset, h->cfg.seed, n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (d_streams) (void)hipFree(d_streams);                   // on the error path too
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(TRL_E_HIP, "trl_env_reset: %s", hipGetErrorString(e)); }
    h->have_reset = true;
    return TRL_OK;
}

int trl_env_get_state(trl_engine *h, trl_game_state *host_out) {
    if (!h || !host_out) return fail(TRL_E_INVALID, "%s", "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(host_out, h->d_states, (size_t)h->cfg.n_games * sizeof(trl_game_state), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return TRL_OK;
}

int trl_env_set_state(trl_engine *h, const trl_game_state *host_in) {
    if (!h || !host_in) return fail(TRL_E_INVALID, "%s", "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(h->d_states, host_in, (size_t)h->cfg.n_games * sizeof(trl_game_state), hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->have_reset = true;
    return TRL_OK;
}

int trl_env_state_dev(trl_engine *h, void **dev_out) {
    if (!h || !dev_out) return fail(TRL_E_INVALID, "%s", "null argument");
    *dev_out = h->d_states;
    return TRL_OK;
}

int trl_env_legal_dev(trl_engine *h, void **dev_counts, void **dev_moves) {
    if (!h) return fail(TRL_E_INVALID, "%s", "null engine");
    if (dev_counts) *dev_counts = h->d_counts;
    if (dev_moves) *dev_moves = h->d_moves;
    return TRL_OK;
}

int trl_env_legal_moves(trl_engine *h, uint16_t *host_counts, uint16_t *host_moves) {
    return trl_env_legal_moves_algo(h, TRL_ALGO_CONVOLUTIONAL, host_counts, host_moves);
}

int trl_env_legal_moves_algo(trl_engine *h, int algo, uint16_t *host_counts, uint16_t *host_moves) {
    if (!h) return fail(TRL_E_INVALID, "%s", "null engine");
    if (algo != TRL_ALGO_CONVOLUTIONAL && algo != TRL_ALGO_BRUTE_FORCE)
        return fail(TRL_E_INVALID, "%s", "Unknown algorithm (the lossy generators 'faster-but-loss' / 'harddrop' are not built)");
    if (!h->have_reset) return fail(TRL_E_STATE, "%s", "trl_env_legal_moves before trl_env_reset / trl_env_set_state");
    HIPCHK(hipSetDevice(h->cfg.device));
    int n = h->cfg.n_games;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    if (algo == TRL_ALGO_BRUTE_FORCE)
        hipLaunchKernelGGL(k_legal_bruteforce, dim3(n), dim3(BLOCK), 0, h->stream, h->d_states, h->d_counts, h->d_moves);
    else
        hipLaunchKernelGGL(k_legal, dim3(n), dim3(BLOCK), 0, h->stream, h->d_states, h->d_counts, h->d_moves);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if (host_counts) HIPCHK(hipMemcpyAsync(host_counts, h->d_counts, (size_t)n * 2, hipMemcpyDeviceToHost, h->stream));
    if (host_moves) HIPCHK(hipMemcpyAsync(host_moves, h->d_moves, (size_t)n * TRL_MAX_LEGAL * 2, hipMemcpyDeviceToHost, h->stream));
    if (host_counts || host_moves) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    }
    return TRL_OK;
}

int trl_env_step_dev(trl_engine *h, const void *dev_moves, int add_bag, int check_legal) {
    if (!h || !dev_moves) return fail(TRL_E_INVALID, "%s", "null argument");
    if (!h->have_reset) return fail(TRL_E_STATE, "%s", "trl_env_step before trl_env_reset / trl_env_set_state");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(k_step, dim3(h->cfg.n_games), dim3(BLOCK), 0, h->stream, h->d_states,
                       (const uint16_t *)dev_moves, h->cfg.seed, add_bag, check_legal);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    return TRL_OK;
}

int trl_env_step(trl_engine *h, const uint16_t *host_moves, int add_bag, int check_legal) {
    if (!h || !host_moves) return fail(TRL_E_INVALID, "%s", "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemcpyAsync(h->d_in_moves, host_moves, (size_t)h->cfg.n_games * 2, hipMemcpyHostToDevice, h->stream));
    int rc = trl_env_step_dev(h, h->d_in_moves, add_bag, check_legal);
    if (rc != TRL_OK) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    return TRL_OK;
}

int trl_env_rollout_random(trl_engine *h, int n_plies, int max_plies, uint64_t *steps_out) {
    if (!h || n_plies < 0 || max_plies <= 0) return fail(TRL_E_INVALID, "%s", "bad argument");
    if (!h->have_reset) return fail(TRL_E_STATE, "%s", "trl_env_rollout_random before trl_env_reset / trl_env_set_state");
    HIPCHK(hipSetDevice(h->cfg.device));
    int pacing = h->cfg.n_games <= 16 * h->n_cu;                            // 16 workgroups of this kernel fit a CU
    int r3 = 4, r2 = 16, r1 = 40;                                           // ranks (of 64) that get priority 3 / 2 / 1
    if (const char *e = getenv("TRL_PACE")) pacing = atoi(e) != 0;          // probe knobs (scripts/rollout_tail_probe.py)
    if (const char *e = getenv("TRL_PACE_R")) sscanf(e, "%d,%d,%d", &r3, &r2, &r1);
    const int rankt = pacing ? ((r3 & 255) | ((r2 & 255) << 8) | ((r1 & 255) << 16)) : 0;
    char *blk = reinterpret_cast<char *>(h->d_steps) + (size_t)h->pp * h->pp_stride;
    char *nxt = reinterpret_cast<char *>(h->d_steps) + (size_t)(h->pp ^ 1) * h->pp_stride;
    h->pp ^= 1;
    // probe knob: extra LDS per workgroup = fewer workgroups per CU at a time (the rest are dispatched as others finish)
    unsigned lds_pad = 0;
    if (const char *e = getenv("TRL_ROLLOUT_LDS_PAD")) lds_pad = (unsigned)atoi(e);
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(k_rollout, dim3(h->cfg.n_games), dim3(BLOCK), lds_pad, h->stream, h->d_states, h->cfg.seed,
                       n_plies, max_plies, reinterpret_cast<unsigned long long *>(blk), reinterpret_cast<unsigned int *>(blk + 16),
                       rankt, reinterpret_cast<unsigned long long *>(nxt), reinterpret_cast<unsigned int *>(nxt + 16));
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    if (steps_out) {
        HIPCHK(hipMemcpyAsync(h->h_steps, blk, sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));   // pinned: no staging
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
        *steps_out = *h->h_steps;
    }
    return TRL_OK;
}

int trl_last_kernel_ms(trl_engine *h, float *ms_out) {
    if (!h || !ms_out) return fail(TRL_E_INVALID, "%s", "null argument");
    HIPCHK(hipEventSynchronize(h->ev1));
    HIPCHK(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    *ms_out = h->last_ms;
    return TRL_OK;
}

#ifdef TRL_SLOT_TIMES
int trl_debug_slot_times(unsigned long long *host_out, int n) {
    HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_slot_t), sizeof(unsigned long long) * 2 * (size_t)n));
    return TRL_OK;
}
#endif
#ifdef TRL_PROFILE
int trl_debug_prof(unsigned long long *host_out, int reset) {
    HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(trl::g_prof), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(trl::g_prof), z, sizeof z)); }
    return TRL_OK;
}
int trl_debug_cprof(unsigned long long *host_out, int reset) {       // closure search (trl_device.h: g_cprof)
    HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(trl::g_cprof), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(trl::g_cprof), z, sizeof z)); }
    return TRL_OK;
}
#endif

// test hook (declared in tests only): integer attack formulas, rows int16[n][8] -> out int16[n][4]
int trl_debug_attack_table(const int16_t *host_rows, int n, int16_t *host_out) {
    int16_t *d_rows = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(&d_rows, (size_t)n * 16);
    if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * 8);
    if (e == hipSuccess) e = hipMemcpy(d_rows, host_rows, (size_t)n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_attack_table, dim3((n + 255) / 256), dim3(256), 0, 0, d_rows, n, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(host_out, d_out, (size_t)n * 8, hipMemcpyDeviceToHost);
    hipFree(d_rows);                                   // (both buffers go whatever happened above)
    hipFree(d_out);
    HIPCHK(e);
    return TRL_OK;
}

}  // extern "C"
