// trl_mcts.hip -- GPU-resident PUCT search + game driver (ai.py:299-659, 1570-1699), gfx950.
//
// One 128-thread workgroup per game slot and tick.  The tree of a slot lives in HBM as
// structure-of-arrays node columns (children of a node are contiguous, so a selection step is
// a coalesced scan of N / Q / P columns reduced by a block arg-max with the reference's
// "last maximal child wins" rule); expanded positions live in a per-slot pool of 320-byte
// states.  Selection, Game.copy + make_move, legal-move search and the net's input features
// are one kernel (k_select); expansion, root noise, backup, FPU refresh and -- on the last
// simulation of a search -- move choice, record, the real make_move and game turnover are the
// other (k_expand).  The net runs between the two on the whole batch of leaves.
//
// Arithmetic mirrors the reference under numpy >= 2 (see oracle/tetris_oracle_mcts.c): float32
// PUCT terms for float32-typed priors (nodes below the root), fp64 at the root and for value
// sums.  This file must be compiled with -ffp-contract=off (no fused multiply-add: the
// reference rounds after every operation).
#include <hip/hip_runtime.h>
#include <math.h>
#include <new>
#include <stdlib.h>
#include "trl_internal.h"

using namespace trl;

static_assert(sizeof(trl_record) == 2392, "trl_record layout");

namespace {

constexpr int DEFAULT_MAX_NODES = 65536;
constexpr int DEFAULT_REC_CAP = 16384;
constexpr int HEAD_MAX_FEATS = 2048;          // widest policy-head input handled in the expand kernel (BaseResNet: 1653)

constexpr int PATH_CAP = 64;                  // deeper paths are backed up by walking the parent links

struct Tree {               // device pointers, column per node field; slot s owns [s * max_nodes, (s + 1) * max_nodes)
    int32_t  *parent, *first, *visit, *state;
    uint16_t *nchild, *move;
    double   *wsum, *q, *prior;
    uint8_t  *pf32, *turn;
};

struct DevMcts {
    trl_mcts_config cfg;
    int n_games, max_nodes, cap_states, rec_cap;
    uint64_t seed;
    Tree t;
    SlotCtl *ctl;
    trl_game_state *pool;       // [n_games][cap_states]
    uint16_t *leaf_moves;       // [n_games][TRL_MAX_LEGAL]
    int32_t *path;              // [n_games][PATH_CAP]: the nodes k_select walked through, for a lane-parallel backup
    trl_record *recs;
    unsigned int *rec_count;
    float *zcache;              // [n_games][cap_states][2 players][zdim]: tower output per expanded position and board
    int zdim;                   // floats per board: 400 for AlphaSame; (own_kernels + 1) * 400 for the BaseResNet family
    unsigned long long *counters;   // 0 simulations, 1 moves, 2 games, 3 dropped, 4 max nodes, 5 overflow, 6 ticks, 7 board-table hits
    // Boards the tower has seen in the slot's current search: [n_games][btab_size] open-addressed words
    //   hash32 (never 0) | (state index | player << 30) << 32,   all zero = empty; cleared when a search begins.
    // A move changes the mover's board only, and how it changes depends on that board and the move alone, so under
    // different replies of the other player the same board comes up again and again: half of the boards a 160-simulation
    // search sends to the tower (three quarters in a 2000-simulation one) have been there before -- their output is in
    // the per-state cache already.  Candidates are confirmed row by row against the stored position (no trust in the hash).
    unsigned long long *btab;
    int btab_size;                  // power of two >= 2 * cap_states; 0: no table
};

// ---- random draws of the search -----------------------------------------------------------------
__device__ __forceinline__ double rng_unif(uint64_t seed, uint32_t stream, uint32_t purpose, uint64_t ctr) {
    uint64_t k = mix64(seed ^ ((uint64_t)stream * 0xD1342543DE82EF95ull));
    uint64_t u = mix64(k ^ ((uint64_t)purpose << 56) ^ ctr);
    return ((double)(u >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double gamma_draw(uint64_t seed, uint32_t stream, uint64_t base_ctr, double a) {
    double d = (a + 1.0) - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (int attempt = 0; attempt < 64; attempt++) {
        uint64_t cc = (base_ctr * 64 + (uint64_t)attempt) * 4;
        double u1 = rng_unif(seed, stream, 5, cc), u2 = rng_unif(seed, stream, 5, cc + 1);
        double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        double u = rng_unif(seed, stream, 5, cc + 2);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) {
            double ub = rng_unif(seed, stream, 5, cc + 3);
            return d * v * exp(log(ub) / a);
        }
    }
    return -1.0;            // 64 rejections in a row (p < 1e-80): the caller uses 0 and counts the search as overflowed
}
// numpy's pairwise float64 sum (np.sum of a contiguous 1-D array, ai.py:600): blocks of at most 128 elements are summed
// with eight running sums, longer arrays are halved (at a multiple of 8) -- two levels of halving cover n <= 512 =
// TRL_MAX_LEGAL, written out so that nothing is called and no array is indexed at run time
__device__ __forceinline__ double np_sum_block(const double *a, int n) {           // n <= 128
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3]; r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; i++) res += a[i];
    return res;
}
__device__ __forceinline__ double np_sum_256(const double *a, int n) {             // n <= 256
    if (n <= 128) return np_sum_block(a, n);
    int n2 = n / 2; n2 -= n2 % 8;
    return np_sum_block(a, n2) + np_sum_block(a + n2, n - n2);
}
__device__ __forceinline__ double np_sum(const double *a, int n) {                 // n <= 512
    if (n <= 128) return np_sum_block(a, n);
    int n2 = n / 2; n2 -= n2 % 8;
    return np_sum_256(a, n2) + np_sum_256(a + n2, n - n2);
}

// ---- shared memory of the two kernels ---------------------------------------------------------------
template <class BLK> struct SelectSharedT {     // BLK = BlockShared ('convolutional') or BfBlockShared ('brute-force')
    BLK b;
    SlotCtl ctl;
    double red_s[2]; int red_k[2];
    int node;
};
using SelectShared = SelectSharedT<BlockShared>;
struct ExpandShared {
    alignas(16) trl_game_state g;
    SlotCtl ctl;
    float  pf[TRL_MAX_LEGAL];
    double pd[TRL_MAX_LEGAL];
    float  xs[HEAD_MAX_FEATS];
    float  zo[400];
    float  hv[20];
    float  red_f[4];
    double dsum;
    int    flag, chosen, rec_idx, end_idx;
};

__device__ __forceinline__ void copy_state(trl_game_state *dst, const trl_game_state *src, int tid) {
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    if (tid < STATE_DW) d[tid] = s[tid];
}

#ifdef TRL_PROFILE
__device__ unsigned long long g_mprof[16];     // [0..4] k_select phases, [8..12] k_expand phases, cycles summed over slots
#define MPROF_T(v) unsigned long long v = __builtin_readcyclecounter()
__device__ unsigned int g_slotprof[4096 * 16];  // the same per slot, for the last launch: the slowest slots set a launch's length
#define MPROF_ADD(i, a, b) do { if (threadIdx.x == 0) { atomicAdd(&g_mprof[i], (unsigned long long)((b) - (a))); \
                                if (blockIdx.x < 4096) g_slotprof[blockIdx.x * 16 + (i)] = (unsigned int)((b) - (a)); } } while (0)
#else
#define MPROF_T(v)
#define MPROF_ADD(i, a, b)
#endif

// ---- tick part A --------------------------------------------------------------------------------------
// do_legal = 0: the legal-move search of the leaf is left to k_leaf_legal, which runs beside the net's tower on a second
// stream (the tower needs the leaf's boards only); leaf_nlegal is filled in there.
// `carry`: the slot's control block as the expand half of the same launch left it (LDS), instead of a store to and a load
// from global memory
template <class SH>
static __device__ __forceinline__ void select_body(SH &sh, const DevMcts &M, trl_game_state *real, float *grids, float *vec,
                                                   uint8_t *need, int do_legal, const SlotCtl *carry = nullptr) {
    const int tid = threadIdx.x, slot = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t nb = (size_t)slot * M.max_nodes;
    trl_game_state *pool = M.pool + (size_t)slot * M.cap_states;
    const trl_mcts_config &cfg = M.cfg;
    MPROF_T(ms0);
    if (tid == 0) sh.ctl = carry ? *carry : M.ctl[slot];
    __syncthreads();
    if (sh.ctl.phase == 2) {                       // idle slot: feed zeros to the net
        for (int i = tid; i < 2 * TRL_GRID_FLOATS; i += BLOCK)
            grids[((size_t)(i / TRL_GRID_FLOATS) * M.n_games + slot) * TRL_GRID_FLOATS + (i % TRL_GRID_FLOATS)] = 0.f;
        for (int i = tid; i < TRL_VEC_FLOATS; i += BLOCK) vec[(size_t)slot * TRL_VEC_FLOATS + i] = 0.f;
        if (need && tid == 0) { need[slot] = 0; need[M.n_games + slot] = 0; }
        return;
    }
    if (sh.ctl.phase == 0) {
        // new search: root = game.copy() with previews restricted to 5 (ai.py:302-313)
        copy_state(&sh.b.g, real + slot, tid);
        __syncthreads();
        if (tid == 0) {
            trl_game_state &g = sh.b.g;
            for (int i = 0; i < 2; i++)
                if (g.p[i].queue_len > TRL_PREVIEWS) { g.p[i].queue_len = TRL_PREVIEWS; g.p[i].queue &= (1ull << (4 * TRL_PREVIEWS)) - 1; }
            SlotCtl &c = sh.ctl;
            c.fast = 0; c.random_move = 0;
            if (cfg.random_start) {                                          // ai.py:1588-1608
                if (g.ply == 0) c.rand_left = -(0.04 * cfg.dir_s) * log(rng_unif(M.seed, g.stream, 6, 0));
                if (c.rand_left > 0) c.random_move = 1;
            }
            if (c.random_move) c.max_iter = 1;                               // fast_config: MAX_ITER = 1, no playout cap
            else if (cfg.training && cfg.use_playout_cap) {                  // ai.py:322-330
                double denom = cfg.playout_cap_chance * (cfg.playout_cap_mult - 1) + 1;
                if (rng_unif(M.seed, g.stream, 3, g.ply) < cfg.playout_cap_chance)
                    c.max_iter = (int)ceil(cfg.playout_cap_mult * (cfg.max_iter / denom));
                else { c.max_iter = (int)floor(cfg.max_iter / denom); c.fast = 1; }
            } else c.max_iter = cfg.max_iter;
            c.iter = 0; c.n_nodes = 1; c.n_states = 1; c.phase = 1; c.overflow = 0; c.leaf_zsrc = -1;
            M.t.parent[nb] = -1; M.t.first[nb] = -1; M.t.nchild[nb] = 0; M.t.move[nb] = TRL_MOVE_NONE;
            M.t.visit[nb] = 0; M.t.wsum[nb] = 0.0; M.t.q[nb] = 0.0; M.t.prior[nb] = 0.0; M.t.state[nb] = 0;
            M.t.pf32[nb] = 0; M.t.turn[nb] = g.turn;
        }
        __syncthreads();
        copy_state(pool, &sh.b.g, tid);
        if (M.btab_size) {
            unsigned long long *bt = M.btab + (size_t)slot * M.btab_size;
            for (int i = tid; i < M.btab_size; i += BLOCK) bt[i] = 0ull;
        }
        __syncthreads();
    }

    // SELECT (ai.py:346-393): walk down by arg-max of Q + U, ties to the last child
    MPROF_T(ms1); MPROF_ADD(0, ms0, ms1);
    int node = 0, depth = 0;
    int par_node = 0, par_first = 0, par_n = 0;
    while (true) {
        if (tid == 0 && depth < PATH_CAP) M.path[(size_t)slot * PATH_CAP + depth] = node;
        depth++;
        int n = M.t.nchild[nb + node];
        if (n == 0) break;
        const int first = M.t.first[nb + node];
        par_node = node; par_first = first; par_n = n;
        const int parent_visits = M.t.visit[nb + node];
        const double sqrt_parent = sqrt((double)parent_visits);
        const double u_scale = cfg.cpuct * sqrt_parent / cfg.dpuct;
        const bool check_forced = cfg.use_forced && cfg.training && node == 0 && !(cfg.use_playout_cap && sh.ctl.fast);
        double best_s = -2.0; int best_k = -1;
        for (int k = tid; k < n; k += BLOCK) {
            const size_t ci = nb + first + k;
            const int vc = M.t.visit[ci];
            const double p = M.t.prior[ci], q = M.t.q[ci];
            double s;
            if (M.t.pf32[ci]) {                    // float32 expression (numpy weak-scalar promotion)
                float pf = (float)p, U;
                if (vc == 0) U = (float)u_scale * pf;
                else U = (((float)cfg.cpuct * pf) * (float)sqrt_parent) / (float)(cfg.dpuct + vc);
                s = (double)((float)q + U);
            } else {
                double U;
                if (vc == 0) U = u_scale * p;
                else U = cfg.cpuct * p * sqrt_parent / (cfg.dpuct + vc);
                s = q + U;
            }
            if (check_forced && vc >= 1) {         // ai.py:377-381
                double n_forced = sqrt(cfg.c_forced * p * parent_visits);
                if (vc < n_forced) s = INFINITY;
            }
            if (s >= best_s) { best_s = s; best_k = k; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            double os = __shfl_down(best_s, d, 64); int ok = __shfl_down(best_k, d, 64);
            if (os > best_s || (os == best_s && ok > best_k)) { best_s = os; best_k = ok; }
        }
        if (lane == 0) { sh.red_s[wave] = best_s; sh.red_k[wave] = best_k; }
        __syncthreads();
        if (tid == 0) {
            int k = sh.red_k[0]; double s = sh.red_s[0];
            if (sh.red_s[1] > s || (sh.red_s[1] == s && sh.red_k[1] > k)) k = sh.red_k[1];
            sh.node = first + k;
        }
        __syncthreads();
        node = sh.node;
        __syncthreads();
    }

    // STEP (ai.py:398-403): node.game = parent.game.copy(); make_move(move, add_bag=False)
    MPROF_T(ms2); MPROF_ADD(1, ms1, ms2);
    int sidx, pstate = -1;
    if (node != 0) {
        const int par = M.t.parent[nb + node];
        pstate = M.t.state[nb + par];
        copy_state(&sh.b.g, pool + M.t.state[nb + par], tid);
        __syncthreads();
        if (tid < 64) {
            uint32_t gctr = real[slot].garb_ctr;               // the reference draws from one global stream
            make_move(sh.b.g, M.seed, M.t.move[nb + node], false, tid, &gctr);
            if (tid == 0) real[slot].garb_ctr = gctr;
        }
        if (tid == 0) {
            int s = M.t.state[nb + node];
            if (s < 0) { s = sh.ctl.n_states++; M.t.state[nb + node] = s; }
            M.t.turn[nb + node] = sh.b.g.turn;
            sh.ctl.leaf_state = s;
        }
        __syncthreads();
        sidx = sh.ctl.leaf_state;
        copy_state(pool + sidx, &sh.b.g, tid);
    } else {
        copy_state(&sh.b.g, pool, tid);
        sidx = 0;
        __syncthreads();
    }

    // leaf kind + legal moves (ai.py:406-417, 472-479)
    MPROF_T(ms3); MPROF_ADD(2, ms2, ms3);
    const trl_game_state &g = sh.b.g;
    int kind, nlegal = 0;
    if (is_terminal(g)) kind = 2;
    else if (no_move(g)) kind = 1;
    else {
        kind = 0;
        if (do_legal) {
            nlegal = block_movegen(sh.b, tid);
            for (int i = tid; i < nlegal; i += BLOCK) M.leaf_moves[(size_t)slot * TRL_MAX_LEGAL + i] = merged_move(sh.b, i);
        } else if (tid == 0) sh.b.overflow = 0;
    }

    // has the tower seen the mover's board in this search?  (wave 0; the other wave goes on to the features)
    int zsrc = -1;
    if (need && M.btab_size && kind != 2 && node != 0 && tid < 64) {
        const int mover = 1 - g.turn;
        const uint32_t myrow = lane < TRL_ROWS ? (uint32_t)g.p[mover].rows[lane < TRL_ROWS ? lane : 0] : 0u;
        uint32_t hsh = (myrow + 1u) * (0x9E3779B1u * (2u * (uint32_t)lane + 1u));
#pragma unroll
        for (int off = 32; off; off >>= 1) hsh += (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ off) << 2, (int)hsh);
        hsh ^= hsh >> 15; hsh *= 0x2C1B3C6Du; hsh ^= hsh >> 12;
        if (hsh == 0u) hsh = 1u;
        unsigned long long *bt = M.btab + (size_t)slot * M.btab_size;
        const unsigned int mask = (unsigned int)M.btab_size - 1u;
        // eight consecutive words in one read (lanes 0..7); never deleted from, so any match in the window counts
        const unsigned long long e = lane < 8 ? bt[(hsh + (unsigned int)lane) & mask] : ~0ull;
        uint64_t cand = __ballot(lane < 8 && (uint32_t)e == hsh);
        const uint64_t empty = __ballot(lane < 8 && e == 0ull);
        while (cand) {
            const int L = __ffsll((unsigned long long)cand) - 1;
            cand &= cand - 1;
            const uint32_t ref = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(e >> 32), L);
            const int cs = (int)(ref & 0x3FFFFFFFu), cp = (int)(ref >> 30) & 1;
            const uint32_t theirs = lane < TRL_ROWS ? (uint32_t)pool[cs].p[cp].rows[lane < TRL_ROWS ? lane : 0] : 0u;
            if (!__ballot(theirs != myrow)) { zsrc = (int)ref; break; }
        }
        if (zsrc < 0 && empty) {
            const int L = __ffsll((unsigned long long)empty) - 1;
            if (lane == L) bt[(hsh + (unsigned int)lane) & mask] = (unsigned long long)hsh | ((unsigned long long)((uint32_t)sidx | ((uint32_t)mover << 30)) << 32);
        }
    }

    // game_to_X (ai.py:1364-1413): side to move first
    MPROF_T(ms4); MPROF_ADD(3, ms3, ms4);
    {
        const trl_player_state &a = g.p[g.turn], &o = g.p[1 - g.turn];
        for (int i = tid; i < 2 * TRL_GRID_FLOATS; i += BLOCK) {
            int who = i / TRL_GRID_FLOATS, cell = i - who * TRL_GRID_FLOATS;
            const trl_player_state &p = who ? o : a;
            int r = cell / TRL_COLS, c = cell - r * TRL_COLS;
            grids[((size_t)who * M.n_games + slot) * TRL_GRID_FLOATS + cell] = (float)((p.rows[r] >> c) & 1);
        }
        float *v = vec + (size_t)slot * TRL_VEC_FLOATS;
        for (int i = tid; i < TRL_VEC_FLOATS; i += BLOCK) {
            float x;
            if (i == 104) x = (float)g.turn;                                  // color of the side to move
            else {
                const trl_player_state &p = (i < 52) ? a : o;
                int j = (i < 52) ? i : i - 52;
                if (j < 49) {                                                 // pieces one-hot [7][7]
                    int row = j / 7, col = j - row * 7;
                    int pc = row == 0 ? (int)p.piece : row == 1 ? (int)p.held
                           : ((row - 2) < (int)p.queue_len ? (int)((p.queue >> (4 * (row - 2))) & 15) : P_NONE);
                    x = (pc == col) ? 1.f : 0.f;
                } else x = (j == 49) ? (float)p.b2b : (j == 50) ? (float)p.combo : (float)p.garbage_len;
            }
            v[i] = x;
        }
    }

    if (tid == 0) {
        SlotCtl &c = sh.ctl;
        c.iter += 1;
        c.leaf = node; c.leaf_kind = kind; c.leaf_nlegal = nlegal; c.leaf_state = sidx; c.leaf_parent_state = pstate;
        c.depth = depth;
        c.leaf_turn = g.turn; c.par_node = par_node; c.par_first = par_first; c.par_n = par_n;
        c.leaf_zsrc = zsrc;
        if (zsrc >= 0) atomicAdd(&M.counters[7], 1ull);
        if (need) {
            // which boards need the tower: a move changes only the mover's board (game.py:66-118), so the side to
            // move at the leaf keeps the tower output its board had in the parent position (trl_mcts_zcache)
            need[slot] = (kind != 2 && node == 0) ? 1 : 0;                    // grids[0]: side to move
            need[M.n_games + slot] = (kind != 2 && zsrc < 0) ? 1 : 0;         // grids[1]: the player who just moved (unless seen: btab)
        }
        if (kind == 2) {
            int w = g.p[0].game_over ? 1 : (g.p[1].game_over ? 0 : -1);      // Game.winner, game.py:217-225
            c.terminal_value = (w == g.turn) ? 1.0 : (w == 1 - g.turn) ? (cfg.use_tanh ? -1.0 : 0.0) : (cfg.use_tanh ? 0.0 : 0.5);
        }
        if (kind == 0 && do_legal && sh.b.overflow) { c.overflow = 1; atomicAdd(&M.counters[5], 1ull); }     // a legal-move cap was hit
        M.ctl[slot] = c;
    }
    MPROF_T(ms5); MPROF_ADD(4, ms4, ms5);
}

__global__ __launch_bounds__(BLOCK) void k_select(DevMcts M, trl_game_state *real, float *grids, float *vec, uint8_t *need,
                                                  int do_legal) {
    __shared__ SelectShared sh;
    select_body(sh, M, real, grids, vec, need, do_legal);
}
// Config.move_algorithm = 'brute-force' (ai.py:84, 414): the leaf's legal moves from the reference's cell-by-cell generator.
// Its own kernel, so that the default one keeps its 10 KB of LDS; not part of the fused expand + select launch.
__global__ __launch_bounds__(BLOCK) void k_select_bf(DevMcts M, trl_game_state *real, float *grids, float *vec, uint8_t *need) {
    __shared__ SelectSharedT<BfBlockShared> sh;
    select_body(sh, M, real, grids, vec, need, 1);
}

// the legal-move search of every slot's current leaf (get_move_matrix + get_move_list, ai.py:414-417), split off k_select:
// 64 VGPRs and 10 KB of LDS per workgroup, so it runs in the registers the tower's two workgroups per CU leave free
__global__ __launch_bounds__(BLOCK, 8) void k_leaf_legal(DevMcts M) {
    __shared__ BlockShared sh;
    const LeafLegalJob J{M.ctl, M.pool, M.leaf_moves, M.counters, M.cap_states, M.n_games};
    leaf_legal_block(J, sh, blockIdx.x, threadIdx.x);
}

// ---- tick part B --------------------------------------------------------------------------------------
// HEAD = false: `policy` is the softmaxed 11 583-vector of every leaf (evaluate(), ai.py:1222-1319).
// HEAD = true : `policy` is the 521-feature vector of every leaf and (head_w, head_b) the policy
//               head; only the logits of the legal moves are formed (one wavefront per move, the
//               weight row read coalesced), p_i = exp(l_i - max_legal).  The softmax denominator
//               cancels in every use the search makes of the priors (ai.py:428-443).
// MODE 0: softmaxed policy given; MODE 1: policy-head input given (HEAD path above);
// MODE 2: AlphaSame heads entirely in this kernel -- `policy` is the tower output z[2][n][400] of the boards
//         k_select flagged, `values` is the packed feature vector vec[n][105]; the kernel completes z from the
//         tower-output cache (see k_zcache), forms the 521-vector (architectures.py:132-140), the opponent-side
//         dense layer, the value head and the legal moves' logits.
struct HeadsDev {
    const float *oside_w, *oside_b;      // [16][400], [16]   (Linear + BatchNorm1d folded)
    const float *v0_w, *v0_b;            // [16][521], [16]   (Linear + BatchNorm1d folded)
    const float *v3_w;                   // [16]
    float v3_b;
    int use_tanh;
    const uint8_t *slot_mask;            // optional: only slots with slot_mask[slot] == mask_want are processed
    int mask_want;                       // (two-network matches: one launch per network, gating.Battle)
};

// NT = threads of the workgroup that take part (128, or 256 in the fused launch: the dense heads and the legal moves'
// policy rows are independent dot products, twice the waves halve their dependent load steps)
template <int MODE, int NT = BLOCK>
static __device__ __forceinline__ void expand_body(ExpandShared &sh, const DevMcts &M, trl_game_state *real, const float *values,
                                                   const float *policy, const float *head_w, const float *head_b, int feat_dim,
                                                   const HeadsDev &H) {
    constexpr bool HEAD = (MODE >= 1);
    constexpr int NW = NT / 64, RPW = 16 / NW;       // waves; rows of a 16-row dense layer per wave
    const int tid = threadIdx.x, slot = blockIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t nb = (size_t)slot * M.max_nodes;
    trl_game_state *pool = M.pool + (size_t)slot * M.cap_states;
    const trl_mcts_config &cfg = M.cfg;
    MPROF_T(me0);
    if (H.slot_mask && H.slot_mask[slot] != (uint8_t)H.mask_want) return;
    if (tid == 0) sh.ctl = M.ctl[slot];
    __syncthreads();
    if (sh.ctl.phase != 1) return;
    if (tid == 0) atomicAdd(&M.counters[0], 1ull);         // simulations = selections whose result was backed up
    const int leaf = sh.ctl.leaf, kind = sh.ctl.leaf_kind;
    const double value_min = cfg.use_tanh ? -1.0 : 0.0;                     // Config.value_min; negate_value below (ai.py:198-216)
    const bool tanh_range = cfg.use_tanh != 0;
    if (MODE == 2 && kind != 2) {
        // complete z from the cache, store the leaf's entry, build x = [z_own, own extras, oside(z_opp), opp extras, colour]
        const float *z = policy, *vecp = values + (size_t)slot * TRL_VEC_FLOATS;
        const int q = sh.ctl.leaf_turn, ps = sh.ctl.leaf_parent_state;        // (= pool[leaf_state].turn)
        float *zc = M.zcache + ((size_t)slot * M.cap_states) * 800;
        float *mine = zc + (size_t)sh.ctl.leaf_state * 800;
        const int zs = sh.ctl.leaf_zsrc;                                      // the mover's board was evaluated before (btab)
        const float *osrc = zs >= 0 ? zc + (size_t)(zs & 0x3FFFFFFF) * 800 + ((zs >> 30) & 1) * 400 : z + ((size_t)M.n_games + slot) * 400;
        for (int i = tid; i < 400; i += NT) {
            float a = (ps < 0) ? z[(size_t)slot * 400 + i] : zc[(size_t)ps * 800 + q * 400 + i];
            float o = osrc[i];
            sh.xs[i] = a; sh.zo[i] = o;
            mine[q * 400 + i] = a; mine[(1 - q) * 400 + i] = o;
        }
        for (int i = tid; i < TRL_VEC_FLOATS; i += NT) sh.xs[i < 52 ? 400 + i : 416 + i] = vecp[i];
        __syncthreads();
        {                                                                     // opponent-side dense 400 -> 16, ReLU
            float zr[7];                                                      // all weight loads of a wave's 8 rows are issued
#pragma unroll                                                                 // before the first use: one L2 round trip
            for (int t = 0; t < 7; t++) zr[t] = (lane + 64 * t < 400) ? sh.zo[lane + 64 * t] : 0.f;
            float acc[RPW];
#pragma unroll
            for (int jj = 0; jj < RPW; jj++) {
                const float *row = H.oside_w + (wave * RPW + jj) * 400;
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 7; t++) a = fmaf((lane + 64 * t < 400) ? row[lane + 64 * t] : 0.f, zr[t], a);
                acc[jj] = a;
            }
#pragma unroll
            for (int jj = 0; jj < RPW; jj++) {
                float a = acc[jj];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
                if (lane == 0) sh.xs[452 + wave * RPW + jj] = fmaxf(a + H.oside_b[wave * RPW + jj], 0.f);
            }
        }
        __syncthreads();
        {                                                                     // value head 521 -> 16, ReLU
            float xr[9];
#pragma unroll
            for (int t = 0; t < 9; t++) xr[t] = (lane + 64 * t < 521) ? sh.xs[lane + 64 * t] : 0.f;
            float acc[RPW];
#pragma unroll
            for (int jj = 0; jj < RPW; jj++) {
                const float *row = H.v0_w + (wave * RPW + jj) * 521;
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 9; t++) a = fmaf((lane + 64 * t < 521) ? row[lane + 64 * t] : 0.f, xr[t], a);
                acc[jj] = a;
            }
#pragma unroll
            for (int jj = 0; jj < RPW; jj++) {
                float a = acc[jj];
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
                if (lane == 0) sh.hv[wave * RPW + jj] = fmaxf(a + H.v0_b[wave * RPW + jj], 0.f);
            }
        }
        __syncthreads();
        if (tid == 0) {
            float v = H.v3_b;
            for (int j = 0; j < 16; j++) v = fmaf(sh.hv[j], H.v3_w[j], v);
            sh.hv[16] = H.use_tanh ? tanhf(v) : 1.0f / (1.0f + expf(-v));
        }
        __syncthreads();
    }
    double value = (kind == 2) ? sh.ctl.terminal_value : (MODE == 2 ? (double)sh.hv[16] : (double)values[slot]);

    // EXPAND (ai.py:411-458)
    MPROF_T(me1); MPROF_ADD(8, me0, me1);
    if (kind == 0) {
        int n = sh.ctl.leaf_nlegal;
        const int first = sh.ctl.n_nodes;
        if (first + n > M.max_nodes) { n = 0; if (tid == 0) { sh.ctl.overflow = 1; atomicAdd(&M.counters[5], 1ull); } }
        const uint16_t *lm = M.leaf_moves + (size_t)slot * TRL_MAX_LEGAL;
        float mymax = 0.f;
        if (HEAD) {
            if (MODE == 1) {
                const float *xf = policy + (size_t)slot * feat_dim;
                for (int i = tid; i < feat_dim; i += NT) sh.xs[i] = xf[i];  // the leaf's feature vector, staged once
                __syncthreads();
            }
            float lmax = -INFINITY;
            if (MODE == 2) {
                float xr[9];
#pragma unroll
                for (int t = 0; t < 9; t++) xr[t] = (lane + 64 * t < 521) ? sh.xs[lane + 64 * t] : 0.f;
                constexpr int RS = 6;                                      // moves per wave and step: 54 loads in flight
                for (int k0 = wave * RS; k0 < n; k0 += NW * RS) {
                    float acc[RS]; int idx[RS];
#pragma unroll
                    for (int u = 0; u < RS; u++) {
                        int k = k0 + u < n ? k0 + u : n - 1;
                        int m = lm[k];
                        idx[u] = (TRL_MOVE_PLANE(m) * TRL_POLICY_ROWS + TRL_MOVE_ROW(m)) * TRL_POLICY_COLS + TRL_MOVE_COL(m);
                    }
#pragma unroll
                    for (int u = 0; u < RS; u++) {
                        const float *row = head_w + (size_t)idx[u] * 521;
                        float a = 0.f;
#pragma unroll
                        for (int t = 0; t < 9; t++) a = fmaf((lane + 64 * t < 521) ? row[lane + 64 * t] : 0.f, xr[t], a);
                        acc[u] = a;
                    }
#pragma unroll
                    for (int u = 0; u < RS; u++) {
                        float a = acc[u];
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
                        a += head_b[idx[u]];
                        if (k0 + u < n) { if (lane == 0) sh.pf[k0 + u] = a; lmax = fmaxf(lmax, a); }
                    }
                }
            } else {
                // any feature width (the BaseResNet family: 1 657): four moves per wave and step share every read of the
                // staged feature vector and keep four times the weight-row loads in flight (one move per step before:
                // k_expand<1> 104 us per tick of 1024 slots)
                for (int k0 = wave * 4; k0 < n; k0 += NW * 4) {
                    const float *row[4]; int idx[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int m = lm[k0 + u < n ? k0 + u : n - 1];
                        idx[u] = (TRL_MOVE_PLANE(m) * TRL_POLICY_ROWS + TRL_MOVE_ROW(m)) * TRL_POLICY_COLS + TRL_MOVE_COL(m);
                        row[u] = head_w + (size_t)idx[u] * feat_dim;
                    }
                    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
                    for (int i = lane; i < feat_dim; i += 64) {
                        const float xv = sh.xs[i];
#pragma unroll
                        for (int u = 0; u < 4; u++) acc[u] = fmaf(row[u][i], xv, acc[u]);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        float a = acc[u];
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) a += __shfl_xor(a, d, 64);
                        a += head_b[idx[u]];
                        if (k0 + u < n) { if (lane == 0) sh.pf[k0 + u] = a; lmax = fmaxf(lmax, a); }
                    }
                }
            }
            if (lane == 0) sh.red_f[wave] = lmax;
            __syncthreads();
            lmax = sh.red_f[0];
#pragma unroll
            for (int w = 1; w < NW; w++) lmax = fmaxf(lmax, sh.red_f[w]);
            // the independent dot products are done: waves beyond the first two end here (a wave that has ended no longer
            // counts at the workgroup's barriers), everything below is a dependent chain that more threads only lengthen
            if (NT > BLOCK && tid >= BLOCK) return;
            __syncthreads();
            for (int k = tid; k < n; k += BLOCK) {
                float p = expf(sh.pf[k] - lmax);
                if (p <= 0.f) p = 1e-25f;
                sh.pf[k] = p;
                mymax = fmaxf(mymax, p);
            }
        } else {
            const float *pol = policy + (size_t)slot * TRL_POLICY_SIZE;
            for (int k = tid; k < n; k += BLOCK) {
                int m = lm[k];
                float p = pol[(TRL_MOVE_PLANE(m) * TRL_POLICY_ROWS + TRL_MOVE_ROW(m)) * TRL_POLICY_COLS + TRL_MOVE_COL(m)];
                if (p <= 0.f) p = 1e-25f;                                          // ai.py:411
                sh.pf[k] = p;
                mymax = fmaxf(mymax, p);
            }
        }
        MPROF_T(mex1); MPROF_ADD(13, me1, mex1);                              // legal-move logits
        const bool root_soft = (leaf == 0) && cfg.use_root_softmax;
        if (root_soft) {                                                       // ai.py:428-434
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) mymax = fmaxf(mymax, __shfl_down(mymax, d, 64));
            if (lane == 0) sh.red_f[wave] = mymax;
            __syncthreads();
            float rmax = sh.red_f[0];
#pragma unroll
            for (int w = 1; w < BLOCK / 64; w++) rmax = fmaxf(rmax, sh.red_f[w]);   // (only the first two waves are left here)
            const double log_max = log((double)rmax);
            const double inv_temp = 1.0 / cfg.root_softmax_temp;
            for (int k = tid; k < n; k += BLOCK) sh.pd[k] = exp((log((double)sh.pf[k]) - log_max) * inv_temp);
        }
        __syncthreads();
        if (tid == 0) {                                                        // sum(policies), in list order
            if (root_soft) { double s = 0.0; for (int k = 0; k < n; k++) s += sh.pd[k]; sh.dsum = s; }
            else { float s = 0.f; for (int k = 0; k < n; k++) s = s + sh.pf[k]; sh.red_f[0] = s; }
        }
        __syncthreads();
        const double q0 = cfg.fpu_reduction ? fmax(value_min, tanh_range ? -value : 1.0 - value)
                                            : fmax(value_min, cfg.fpu_value);                                  // ai.py:447-453
        const double dsum = root_soft ? sh.dsum : 0.0;
        const float fsum = sh.red_f[0];
        for (int k = tid; k < n; k += BLOCK) {
            const size_t ci = nb + first + k;
            M.t.parent[ci] = leaf; M.t.first[ci] = -1; M.t.nchild[ci] = 0; M.t.move[ci] = lm[k];
            M.t.visit[ci] = 0; M.t.wsum[ci] = 0.0; M.t.q[ci] = q0; M.t.state[ci] = -1; M.t.turn[ci] = 0;
            if (root_soft) { M.t.prior[ci] = sh.pd[k] / dsum; M.t.pf32[ci] = 0; }
            else { M.t.prior[ci] = (double)(sh.pf[k] / fsum); M.t.pf32[ci] = 1; }
        }
        MPROF_T(mex2); MPROF_ADD(14, mex1, mex2);                             // normalisation + child initialisation
        if (tid == 0 && n > 0) {
            M.t.first[nb + leaf] = first; M.t.nchild[nb + leaf] = (uint16_t)n;
            sh.ctl.n_nodes = first + n;
            atomicMax(&M.counters[4], (unsigned long long)(first + n));
        }
        __syncthreads();
    }

    if (NT > BLOCK && tid >= BLOCK) return;
    // NOISE (ai.py:482-499): un-normalised gamma draws on the root's children
    MPROF_T(me2); MPROF_ADD(9, me1, me2);
    if (cfg.training && !sh.ctl.fast && cfg.use_dirichlet && leaf == 0 && !sh.ctl.random_move) {
        const int n = M.t.nchild[nb], first = M.t.first[nb];
        if (n > 0) {
            double a = cfg.dir_alpha;
            if (cfg.use_dir_s) a *= cfg.dir_s / n;
            const uint32_t stream = real[slot].stream, ply = real[slot].ply;
            for (int k = tid; k < n; k += BLOCK) {
                const size_t ci = nb + first + k;
                double noise = gamma_draw(M.seed, stream, (uint64_t)ply * 512 + (uint64_t)k, a);
                if (noise < 0.0) { noise = 0.0; atomicAdd(&M.counters[5], 1ull); }      // never seen; not silent if it happens
                double p = M.t.prior[ci];
                if (M.t.pf32[ci]) p = (double)((float)p * (float)(1 - cfg.dir_expl)) + noise * cfg.dir_expl;
                else p = p * (1 - cfg.dir_expl) + noise * cfg.dir_expl;
                M.t.prior[ci] = p; M.t.pf32[ci] = 0;
            }
        }
        __syncthreads();
    }

    // BACKUP (ai.py:504-533) and FPU refresh of the unvisited siblings (ai.py:536-565)
    MPROF_T(me3); MPROF_ADD(10, me2, me3);
    {
        // every node of the path gets the same two numbers, so the walk up the parent links (one dependent global
        // load per level) is replaced by one lane per level over the path k_select recorded
        const double bval = tanh_range ? -value : 1.0 - value;
        const double pos = bval, neg = tanh_range ? -bval : 1.0 - bval;
        const int final_turn = sh.ctl.leaf_turn;                               // (= M.t.turn[nb + leaf])
        const int depth = sh.ctl.depth;
        if (depth <= PATH_CAP) {
            if (tid < depth) {
                const size_t i = nb + M.path[(size_t)slot * PATH_CAP + tid];
                const int vc = M.t.visit[i] + 1;
                const double w = M.t.wsum[i] + ((M.t.turn[i] == final_turn) ? pos : neg);
                M.t.visit[i] = vc; M.t.wsum[i] = w; M.t.q[i] = w / vc;
            }
        } else if (tid == 0) {
            for (int n = leaf;; n = M.t.parent[nb + n]) {
                const size_t i = nb + n;
                int vc = M.t.visit[i] + 1;
                double w = M.t.wsum[i] + ((M.t.turn[i] == final_turn) ? pos : neg);
                M.t.visit[i] = vc; M.t.wsum[i] = w; M.t.q[i] = w / vc;
                if (n == 0) break;
            }
        }
        __threadfence_block();
        __syncthreads();
        // FPU refresh: the parent's unvisited children restart from (1 - Q_parent) - reduction * sqrt(sum of the visited
        // children's priors), the sum taken in float32 and in child order (ai.py:536-565)
        const int par = (leaf != 0 && cfg.fpu_reduction) ? sh.ctl.par_node : 0;   // (= M.t.parent[nb + leaf])
        if (par != 0) {
            const int n = sh.ctl.par_n, first = sh.ctl.par_first;              // (= nchild / first of the parent)
            for (int k = tid; k < n; k += BLOCK)
                sh.pf[k] = (M.t.visit[nb + first + k] > 0) ? (float)M.t.prior[nb + first + k] : -1.f;   // priors are > 0
            __syncthreads();
            if (tid == 0) {
                float explored = 0.f; bool any = false;
                for (int k = 0; k < n; k++)
                    if (sh.pf[k] >= 0.f) { explored = any ? explored + sh.pf[k] : sh.pf[k]; any = true; }
                const double qp = M.t.q[nb + par];
                double fpu = (tanh_range ? -qp : 1.0 - qp) - cfg.fpu_value * sqrt((double)explored);
                if (!(fpu > value_min)) fpu = value_min;
                sh.dsum = fpu;
            }
            __syncthreads();
            const double fpu = sh.dsum;
            for (int k = tid; k < n; k += BLOCK)
                if (sh.pf[k] < 0.f) M.t.q[nb + first + k] = fpu;
        }
        __syncthreads();
    }

    // end of the search: choose, record, play (ai.py:568-614, 1610-1695)
    MPROF_T(me4); MPROF_ADD(11, me3, me4);
    if (sh.ctl.iter >= sh.ctl.max_iter) {
        const int n = M.t.nchild[nb], first = M.t.first[nb];
        copy_state(&sh.g, real + slot, tid);
        for (int k = tid; k < n; k += BLOCK) sh.pd[k] = (double)M.t.visit[nb + first + k];
        __syncthreads();
        if (tid == 0) {
            int sel = 0;
            const double temp = cfg.training ? cfg.temperature : 0.0;
            if (n > 0 && sh.ctl.random_move) {
                // pick_random_move_by_policy (ai.py:999-1014): random.choices(moves, priors); float32-typed priors (no root
                // softmax) accumulate in float32 as numpy scalars do
                const bool f32 = M.t.pf32[nb + first] != 0;
                float af = 0.f; double ad = 0.0;
                for (int k = 0; k < n; k++) {
                    const double pk = M.t.prior[nb + first + k];
                    if (f32) { af = (k == 0) ? (float)pk : af + (float)pk; sh.pd[k] = (double)af; }
                    else { ad = (k == 0) ? pk : ad + pk; sh.pd[k] = ad; }
                }
                const double total = sh.pd[n - 1] + 0.0;
                const double x = rng_unif(M.seed, sh.g.stream, 4, sh.g.ply) * total;
                int lo = 0, hi = n - 1;
                while (lo < hi) { int mid = (lo + hi) / 2; if (x < sh.pd[mid]) hi = mid; else lo = mid + 1; }
                sel = lo;
                sh.ctl.rand_left -= 1.0;
            } else if (n > 0) {
                if (temp == 0.0) {                                             // np.argmax: first maximum
                    double best = -1.0;
                    for (int k = 0; k < n; k++) if (sh.pd[k] > best) { best = sh.pd[k]; sel = k; }
                } else {                                                       // N^(1/T), normalise, random.choices
                    for (int k = 0; k < n; k++) sh.pd[k] = pow(sh.pd[k], 1.0 / temp);
                    double s = np_sum(sh.pd, n);
                    double acc = 0.0;
                    for (int k = 0; k < n; k++) { double pr = sh.pd[k] / s; acc = (k == 0) ? pr : acc + pr; sh.pd[k] = acc; }
                    double total = sh.pd[n - 1] + 0.0;
                    double x = rng_unif(M.seed, sh.g.stream, 4, sh.g.ply) * total;
                    int lo = 0, hi = n - 1;
                    while (lo < hi) { int mid = (lo + hi) / 2; if (x < sh.pd[mid]) hi = mid; else lo = mid + 1; }
                    sel = lo;
                }
            }
            sh.chosen = n > 0 ? (int)M.t.move[nb + first + sel] : -1;
            // policy-target pruning AFTER the move is chosen (ai.py:616-648): only the recorded visit counts change
            if (cfg.use_forced && cfg.training && !sh.ctl.fast && n > 0 && !sh.ctl.random_move) {
                int max_n = 0, max_k = 0;
                for (int k = 0; k < n; k++) { int v = M.t.visit[nb + first + k]; if (v >= max_n) { max_n = v; max_k = k; } }   // last maximum
                const double rootN = (double)M.t.visit[nb], sq = sqrt(rootN);
                const size_t mi = nb + first + max_k;
                const double most = M.t.q[mi] + cfg.cpuct * M.t.prior[mi] * sq / (cfg.dpuct + M.t.visit[mi]);
                for (int k = 0; k < n; k++) {
                    if (k == max_k) continue;
                    const size_t ci = nb + first + k;
                    int vc = M.t.visit[ci];
                    if (vc > 0) {
                        const double pk = M.t.prior[ci];
                        const double n_forced = sqrt(cfg.c_forced * pk * rootN);
                        // the reference puts the ROOT's N in the denominator (ai.py:640), so the bound is constant
                        const double minus = M.t.q[ci] + cfg.cpuct * pk * sq / (cfg.dpuct + rootN);
                        int count = 0;
                        for (;;) {
                            if (vc == 1) { vc = 0; break; }
                            if (count < n_forced && minus < most) { count += 1; vc -= 1; }
                            else break;
                        }
                        M.t.visit[ci] = vc;
                    }
                }
            }
            unsigned int idx = atomicAdd(M.rec_count, 1u);
            sh.rec_idx = idx < (unsigned)M.rec_cap ? (int)idx : -1;
            if (sh.rec_idx < 0) atomicAdd(&M.counters[3], 1ull);
        }
        __syncthreads();
        if (sh.rec_idx >= 0) {                                                 // the position searched + (move, N) of the root children
            trl_record *r = M.recs + sh.rec_idx;
            copy_state(&r->state, &sh.g, tid);
            for (int k = tid; k < TRL_MAX_LEGAL; k += BLOCK) {
                r->moves[k] = k < n ? M.t.move[nb + first + k] : (uint16_t)0;
                r->visits[k] = k < n ? (uint16_t)M.t.visit[nb + first + k] : (uint16_t)0;
            }
            if (tid == 0) {
                r->slot = slot; r->episode = sh.g.episode; r->ply = sh.g.ply; r->n_children = (uint16_t)n;
                r->kind = TRL_REC_MOVE; r->winner = -1; r->chosen = (uint16_t)sh.chosen;
                r->saved = (uint8_t)(((!sh.ctl.fast) || cfg.save_all) && !sh.ctl.random_move);
                r->random_move = (uint8_t)sh.ctl.random_move;
            }
        }
        __syncthreads();
        if (tid < 64 && sh.chosen >= 0) make_move(sh.g, M.seed, sh.chosen, true, tid);   // game.make_move(move), ai.py:1668
        if (tid == 0) {
            trl_game_state &g = sh.g;
            bool over = true;
            if (sh.chosen >= 0) {
                // while not terminal and len(history.states) < MAX_MOVES  (history: 1 + player-1 placements)
                over = is_terminal(g) || (1 + g.p[1].pieces) >= TRL_MAX_MOVES;
            }
            atomicAdd(&M.counters[1], 1ull);
            sh.end_idx = -1; sh.flag = over ? 1 : 0;
            if (over) {
                unsigned int idx = atomicAdd(M.rec_count, 1u);
                if (idx < (unsigned)M.rec_cap) sh.end_idx = (int)idx; else atomicAdd(&M.counters[3], 1ull);
                atomicAdd(&M.counters[2], 1ull);
            }
        }
        __syncthreads();
        if (sh.flag) {
            if (sh.end_idx >= 0) {
                trl_record *r = M.recs + sh.end_idx;
                copy_state(&r->state, &sh.g, tid);
                if (tid == 0) {
                    const trl_game_state &g = sh.g;
                    r->slot = slot; r->episode = g.episode; r->ply = g.ply; r->n_children = 0; r->kind = TRL_REC_GAME_END;
                    r->winner = (int8_t)(g.p[0].game_over ? 1 : (g.p[1].game_over ? 0 : -1));
                    r->chosen = 0; r->saved = 0; r->random_move = 0;
                }
            }
            __syncthreads();
            if (tid == 0) {
                const int quota = cfg.episodes_base + (slot < cfg.episodes_extra ? 1 : 0);      // 0: no limit
                const bool quota_met = quota > 0 && (int)sh.g.episode + 1 >= quota;     // first: a quota of exactly 256 ends cleanly
                if (cfg.auto_reset && !quota_met && sh.g.episode + 1 >= 256u) {
                    // the next episode's stream (+ 2^24 per episode) would wrap onto this slot's first one and replay its games
                    sh.ctl.overflow = 1; atomicAdd(&M.counters[5], 1ull);
                    sh.ctl.phase = 2;
                } else if (cfg.auto_reset && !quota_met) {
                    trl_game_state &g = sh.g;
                    uint32_t stream = g.stream + 0x01000000u, episode = g.episode + 1;
                    int ruleset = g.ruleset; uint8_t flags = g.flags;
                    game_setup(g, ruleset, M.seed, stream);
                    g.episode = episode; g.flags = flags;
                    sh.ctl.phase = 0;
                } else sh.ctl.phase = 2;
            }
        } else if (tid == 0) sh.ctl.phase = 0;
        __syncthreads();
        copy_state(real + slot, &sh.g, tid);
    }
    if (tid == 0) { M.ctl[slot] = sh.ctl; if (slot == 0) atomicAdd(&M.counters[6], 1ull); }
    MPROF_T(me5); MPROF_ADD(12, me4, me5);
}

template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_expand(DevMcts M, trl_game_state *real, const float *values, const float *policy,
                                                  const float *head_w, const float *head_b, int feat_dim, HeadsDev H) {
    __shared__ ExpandShared sh;
    expand_body<MODE>(sh, M, real, values, policy, head_w, head_b, feat_dim, H);
}

#ifndef TRL_SEARCH_PRIO
#define TRL_SEARCH_PRIO 3
#endif
constexpr int SEARCH_PRIO = TRL_SEARCH_PRIO;

// Tick part B of one simulation and part A of the next in one launch (the production tick with an AlphaSame net): a slot
// that is done with its expand goes on to its next selection instead of waiting for the slowest expand of the launch, and
// the launch ends with the slowest expand + select of one slot, not with the slowest expand plus the slowest select.
// `vec` is read by the expand (the leaf's feature vector) before the same workgroup's select overwrites it; `grids` were
// consumed by the tower that ran before this launch.
constexpr int FUSED_THREADS = 256;        // four waves for the expand half, the first two go on to the selection
template <bool DO_LEGAL>      // false: the leaf's legal-move search runs in the tower launch that follows (LeafLegalJob)
__global__ __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_expand_select(DevMcts M, trl_game_state *real, float *vec, const float *z,
                                                         const float *head_w, const float *head_b, HeadsDev H, float *grids,
                                                         uint8_t *need) {
    __shared__ union U { ExpandShared e; SelectShared s; __device__ U() {} } sh;
    __shared__ SlotCtl carry;                      // the slot's control block from the expand half to the select half
    // latency-bound (two waves per slot, long dependent chains): when the other slot group's tower shares the SIMD, these
    // waves go first -- they issue rarely, so the tower loses little and the search chain of the tick gets shorter
    __builtin_amdgcn_s_setprio(SEARCH_PRIO);
    expand_body<2, FUSED_THREADS>(sh.e, M, real, vec, z, head_w, head_b, 521, H);
    if (threadIdx.x >= BLOCK) return;              // (slots that were not expanding: the extra waves end here)
    if (threadIdx.x == 0) carry = sh.e.ctl;        // as stored to M.ctl[slot] (or as loaded, for a slot that was not searching)
    __syncthreads();                               // the tree is in memory; the LDS block changes hands
    select_body(sh.s, M, real, grids, vec, need, DO_LEGAL ? 1 : 0, &carry);
}

// tower-output cache.  z: [2][n][400] as written by the tower for the boards flagged by k_select.  Afterwards
// z[0] / z[1] hold the outputs for the side to move / the other player of every evaluated leaf, and the
// leaf's cache entry holds both (indexed by absolute player).
__global__ __launch_bounds__(128) void k_zcache(DevMcts M, float *z) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    const SlotCtl c = M.ctl[slot];
    if (c.phase != 1 || c.leaf_kind == 2) return;
    const trl_game_state *pool = M.pool + (size_t)slot * M.cap_states;
    const int q = pool[c.leaf_state].turn;                       // side to move at the leaf
    const int zd = M.zdim;
    float *zc = M.zcache + ((size_t)slot * M.cap_states) * 2 * zd;
    float *mine = zc + (size_t)c.leaf_state * 2 * zd;
    float *za = z + (size_t)slot * zd, *zo = z + ((size_t)M.n_games + slot) * zd;
    const int zs = c.leaf_zsrc;
    const float *osrc = zs >= 0 ? zc + (size_t)(zs & 0x3FFFFFFF) * 2 * zd + ((zs >> 30) & 1) * zd : zo;
    for (int i = tid; i < zd; i += 128) {
        float a = (c.leaf_parent_state < 0) ? za[i] : zc[(size_t)c.leaf_parent_state * 2 * zd + q * zd + i];
        float o = osrc[i];
        zo[i] = o;
        za[i] = a;
        mine[q * zd + i] = a;
        mine[(1 - q) * zd + i] = o;
    }
}

// gather of the current leaf states (tests / host-side evaluators)
__global__ void k_leaf_gather(DevMcts M, trl_game_state *out, int32_t *info) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    const SlotCtl c = M.ctl[slot];
    copy_state(out + slot, M.pool + (size_t)slot * M.cap_states + (c.phase == 1 ? c.leaf_state : 0), tid);
    if (tid == 0) { info[4 * slot] = c.iter; info[4 * slot + 1] = c.max_iter; info[4 * slot + 2] = c.phase == 1 ? c.leaf_kind : 3; info[4 * slot + 3] = c.leaf_nlegal; }
}
__global__ void k_root_children(DevMcts M, int32_t *counts, uint16_t *moves, int32_t *visits, double *q, double *p) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    const size_t nb = (size_t)slot * M.max_nodes;
    const int n = M.t.nchild[nb], first = M.t.first[nb];
    if (tid == 0) counts[slot] = n;
    for (int k = tid; k < n; k += blockDim.x) {
        size_t o = (size_t)slot * TRL_MAX_LEGAL + k, ci = nb + first + k;
        moves[o] = M.t.move[ci]; visits[o] = M.t.visit[ci]; q[o] = M.t.q[ci]; p[o] = M.t.prior[ci];
    }
}

// deterministic stub evaluator (tests): same function as oracle/tetris_oracle_mcts.c to_stub_eval_state
__global__ __launch_bounds__(256) void k_stub_eval(DevMcts M, float *values, float *policy, uint64_t salt) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    const SlotCtl c = M.ctl[slot];
    const trl_game_state *s = M.pool + (size_t)slot * M.cap_states + (c.phase == 1 ? c.leaf_state : 0);
    uint64_t h = 0x243F6A8885A308D3ull;
    for (int i = 0; i < 2; i++) {
        const uint64_t *w = reinterpret_cast<const uint64_t *>(&s->p[i]);
        for (int k = 0; k < 15; k++) h = mix64(h ^ w[k]);
        const trl_player_state &p = s->p[i];
        uint64_t meta = (uint64_t)(uint16_t)p.b2b | ((uint64_t)(uint16_t)p.combo << 16) | ((uint64_t)p.b2b_level << 32)
                      | ((uint64_t)p.piece << 40) | ((uint64_t)p.held << 44) | ((uint64_t)p.queue_len << 48)
                      | ((uint64_t)p.garbage_len << 53) | ((uint64_t)p.game_over << 60);
        h = mix64(h ^ meta);
    }
    h = mix64(h ^ s->turn);
    if (salt) h = mix64(h ^ salt);
    if (tid == 0) {
        const float v = (float)(mix64(h) >> 40) * (1.0f / 16777216.0f);
        values[slot] = M.cfg.use_tanh ? 2.0f * v - 1.0f : v;                // tanh-range stub (oracle: to_stub_eval_tanh)
    }
    for (int i = tid; i < TRL_POLICY_SIZE; i += blockDim.x) {
        uint64_t k = mix64(h ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull));
        float u = (float)(k >> 40) * (1.0f / 16777216.0f);
        policy[(size_t)slot * TRL_POLICY_SIZE + i] = (k & 15) == 0 ? 0.0f : u * u;
    }
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------
struct trl_mcts {
    DevMcts d;
    uint64_t ticks;
    trl_game_state *d_leaf_tmp;
    int32_t *d_info_tmp;
    void *d_root_tmp;
    int fork_legal;                 // trl_mcts_fork_legal: the leaf's legal-move search runs on side_stream beside the tower
    int legal_in_tower;             // trl_mcts_legal_in_tower: ... as extra workgroups of the tower launch (trl_mcts_tower16_legal)
    bool legal_owed;                // a selection without its legal-move search has been queued: the tower launch must follow
    bool fork_pending;              // k_leaf_legal is in flight: the next expand joins it
    hipStream_t side_stream;
    hipEvent_t ev_fork, ev_join;
};

static void free_dev(DevMcts &d) {
    hipFree(d.t.parent); hipFree(d.t.first); hipFree(d.t.visit); hipFree(d.t.state); hipFree(d.t.nchild); hipFree(d.t.move);
    hipFree(d.t.wsum); hipFree(d.t.q); hipFree(d.t.prior); hipFree(d.t.pf32); hipFree(d.t.turn);
    hipFree(d.ctl); hipFree(d.pool); hipFree(d.leaf_moves); hipFree(d.path); hipFree(d.recs); hipFree(d.rec_count); hipFree(d.counters);
    hipFree(d.zcache); hipFree(d.btab);
}

void trl_mcts_free_internal(trl_engine *h) {
    if (!h || !h->mcts) return;
    free_dev(h->mcts->d);
    hipFree(h->mcts->d_leaf_tmp); hipFree(h->mcts->d_info_tmp); hipFree(h->mcts->d_root_tmp);
    if (h->mcts->side_stream) { hipStreamSynchronize(h->mcts->side_stream); hipStreamDestroy(h->mcts->side_stream); }
    if (h->mcts->ev_fork) hipEventDestroy(h->mcts->ev_fork);
    if (h->mcts->ev_join) hipEventDestroy(h->mcts->ev_join);
    delete h->mcts;
    h->mcts = nullptr;
}

static int mcts_alloc(trl_engine *h, trl_mcts *m) {
    DevMcts &d = m->d;
    size_t nn = (size_t)d.n_games * d.max_nodes, n = (size_t)d.n_games;
    HIPCHK(hipMalloc(&d.t.parent, nn * 4)); HIPCHK(hipMalloc(&d.t.first, nn * 4)); HIPCHK(hipMalloc(&d.t.visit, nn * 4));
    HIPCHK(hipMalloc(&d.t.state, nn * 4)); HIPCHK(hipMalloc(&d.t.nchild, nn * 2)); HIPCHK(hipMalloc(&d.t.move, nn * 2));
    HIPCHK(hipMalloc(&d.t.wsum, nn * 8)); HIPCHK(hipMalloc(&d.t.q, nn * 8)); HIPCHK(hipMalloc(&d.t.prior, nn * 8));
    HIPCHK(hipMalloc(&d.t.pf32, nn)); HIPCHK(hipMalloc(&d.t.turn, nn));
    HIPCHK(hipMalloc(&d.ctl, n * sizeof(SlotCtl)));
    HIPCHK(hipMalloc(&d.pool, n * d.cap_states * sizeof(trl_game_state)));
    HIPCHK(hipMalloc(&d.leaf_moves, n * TRL_MAX_LEGAL * 2));
    HIPCHK(hipMalloc(&d.path, n * PATH_CAP * sizeof(int32_t)));
    d.zdim = 400;
    HIPCHK(hipMalloc(&d.zcache, n * d.cap_states * 2 * d.zdim * sizeof(float)));
    if (d.btab_size) {
        HIPCHK(hipMalloc(&d.btab, n * (size_t)d.btab_size * sizeof(unsigned long long)));
        HIPCHK(hipMemsetAsync(d.btab, 0, n * (size_t)d.btab_size * sizeof(unsigned long long), h->stream));
    }
    HIPCHK(hipMalloc(&d.recs, (size_t)d.rec_cap * sizeof(trl_record)));
    HIPCHK(hipMalloc(&d.rec_count, 4));
    HIPCHK(hipMalloc(&d.counters, 8 * 8));
    HIPCHK(hipMalloc(&m->d_leaf_tmp, n * sizeof(trl_game_state)));
    HIPCHK(hipMalloc(&m->d_info_tmp, n * 16));
    HIPCHK(hipMalloc(&m->d_root_tmp, n * (4 + TRL_MAX_LEGAL * (2 + 4 + 8 + 8))));
    HIPCHK(hipMemsetAsync(d.ctl, 0, n * sizeof(SlotCtl), h->stream));            // phase 0: start a search
    HIPCHK(hipMemsetAsync(d.t.nchild, 0, nn * 2, h->stream));
    HIPCHK(hipMemsetAsync(d.rec_count, 0, 4, h->stream));
    HIPCHK(hipMemsetAsync(d.counters, 0, 64, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return TRL_OK;
}

extern "C" {

int trl_mcts_create(trl_engine *h, const trl_mcts_config *cfg) {
    if (!h || !cfg || cfg->max_iter <= 0) return fail(TRL_E_INVALID, "trl_mcts_create: %s", "bad argument");
    if (cfg->move_algorithm != TRL_ALGO_CONVOLUTIONAL && cfg->move_algorithm != TRL_ALGO_BRUTE_FORCE)
        return fail(TRL_E_INVALID, "trl_mcts_create: %s", "Unknown algorithm (the lossy generators 'faster-but-loss' / 'harddrop' are not built)");
    if (!h->have_reset) return fail(TRL_E_STATE, "%s", "trl_mcts_create before trl_env_reset / trl_env_set_state");
    HIPCHK(hipSetDevice(h->cfg.device));
    trl_mcts_free_internal(h);
    trl_mcts *m = new (std::nothrow) trl_mcts();
    if (!m) return fail(TRL_E_INVALID, "%s", "out of host memory");
    memset(m, 0, sizeof *m);
    DevMcts &d = m->d;
    d.cfg = *cfg;
    d.n_games = h->cfg.n_games;
    d.seed = h->cfg.seed;
    d.rec_cap = cfg->record_capacity > 0 ? cfg->record_capacity : DEFAULT_REC_CAP;
    int longest = cfg->max_iter;
    if (cfg->training && cfg->use_playout_cap) {
        double denom = cfg->playout_cap_chance * (cfg->playout_cap_mult - 1) + 1;
        longest = (int)ceil(cfg->playout_cap_mult * (cfg->max_iter / denom));
        if (longest < cfg->max_iter) longest = cfg->max_iter;
    }
    if (longest > 65535) {                         // trl_record.visits is uint16: the longest search must fit
        delete m;
        return fail(TRL_E_INVALID, "trl_mcts_create: %s", "searches longer than 65535 simulations do not fit the record's visit counts");
    }
    if (cfg->episodes_base < 0 || cfg->episodes_extra < 0 || cfg->episodes_base + (cfg->episodes_extra > 0) > 256) {
        delete m;
        return fail(TRL_E_INVALID, "trl_mcts_create: %s", "at most 256 episodes per slot (RNG stream = slot stream + episode * 2^24)");
    }
    d.cap_states = longest + 2;
    d.btab_size = 0;
    {   // TRL_BOARD_TABLE=0: every leaf's board goes to the tower (A/B runs and the tests that compare the two)
        const char *e = getenv("TRL_BOARD_TABLE");
        if (!e || atoi(e) != 0) { d.btab_size = 64; while (d.btab_size < 2 * d.cap_states) d.btab_size <<= 1; }
    }
    // default tree capacity: one expansion per simulation, ~50 children each (measured: 20.5 k nodes after 400
    // simulations), with headroom; never below 65 536
    d.max_nodes = cfg->max_nodes > 0 ? cfg->max_nodes : (longest * 96 > DEFAULT_MAX_NODES ? longest * 96 : DEFAULT_MAX_NODES);
    h->mcts = m;                                   // owned by the engine from here on: a failed allocation below
    int rc = mcts_alloc(h, m);                     // (a tree larger than the free HBM) is released, not leaked
    if (rc != TRL_OK) { trl_mcts_free_internal(h); return rc; }
    return TRL_OK;
}

// every slot starts a new search at its next selection (the state after trl_mcts_create): for callers that replace the
// game states (trl_env_set_state / _reset / _step) while a selection made by trl_mcts_expand_alphasame_select is pending
int trl_mcts_restart(trl_engine *h) {
    if (!h || !h->mcts) return fail(TRL_E_INVALID, "trl_mcts_restart: %s", "no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemsetAsync(h->mcts->d.ctl, 0, (size_t)h->cfg.n_games * sizeof(SlotCtl), h->stream));
    h->mcts->legal_owed = false;                    // the pending selection is dropped with everything else
    return TRL_OK;
}

int trl_mcts_select(trl_engine *h, float *dev_grids, float *dev_vec) {
    return trl_mcts_select_need(h, dev_grids, dev_vec, nullptr);
}

int trl_mcts_select_need(trl_engine *h, float *dev_grids, float *dev_vec, uint8_t *dev_need) {
    if (!h || !h->mcts || !dev_grids || !dev_vec) return fail(TRL_E_INVALID, "trl_mcts_select: %s", "null argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    trl_mcts *m = h->mcts;
    const int fork = m->fork_legal && dev_need;         // only the tower-need flow has something to overlap with
    const int in_tower = m->legal_in_tower && dev_need;
    if (in_tower && m->legal_owed) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: a selection is waiting for trl_mcts_tower16_legal");
    if (m->d.cfg.move_algorithm == TRL_ALGO_BRUTE_FORCE) {
        if (fork || in_tower) return fail(TRL_E_STATE, "%s", "trl_mcts_fork_legal / trl_mcts_legal_in_tower: not with move_algorithm = brute-force");
        hipLaunchKernelGGL(k_select_bf, dim3(h->cfg.n_games), dim3(BLOCK), 0, h->stream, m->d, h->d_states, dev_grids, dev_vec, dev_need);
        HIPCHK(hipGetLastError());
        return TRL_OK;
    }
    hipLaunchKernelGGL(k_select, dim3(h->cfg.n_games), dim3(BLOCK), 0, h->stream, m->d, h->d_states, dev_grids, dev_vec, dev_need,
                       (fork || in_tower) ? 0 : 1);
    HIPCHK(hipGetLastError());
    if (in_tower) m->legal_owed = true;
    if (fork) {
        // fork: side_stream waits for k_select, runs k_leaf_legal; the next expand call joins (mcts_join).  Under stream
        // capture (trl_graph_begin) the two event waits become the graph's fork and join edges.
        HIPCHK(hipEventRecord(m->ev_fork, h->stream));
        HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
        hipLaunchKernelGGL(k_leaf_legal, dim3(h->cfg.n_games), dim3(BLOCK), 0, m->side_stream, m->d);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
        m->fork_pending = true;
    }
    return TRL_OK;
}

static int mcts_join(trl_engine *h) {
    trl_mcts *m = h->mcts;
    if (m->fork_pending) {
        HIPCHK(hipStreamWaitEvent(h->stream, m->ev_join, 0));
        m->fork_pending = false;
    }
    return TRL_OK;
}

int trl_mcts_fork_legal(trl_engine *h, int on) {
    if (!h || !h->mcts) return fail(TRL_E_INVALID, "trl_mcts_fork_legal: %s", "no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    trl_mcts *m = h->mcts;
    if (on && !m->side_stream) {
        HIPCHK(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    }
    if (on && m->legal_in_tower) return fail(TRL_E_STATE, "%s", "trl_mcts_fork_legal: not with trl_mcts_legal_in_tower");
    m->fork_legal = on ? 1 : 0;
    return TRL_OK;
}

int trl_mcts_legal_in_tower(trl_engine *h, int on) {
    if (!h || !h->mcts) return fail(TRL_E_INVALID, "trl_mcts_legal_in_tower: %s", "no search attached");
    trl_mcts *m = h->mcts;
    if (on && m->fork_legal) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: not with trl_mcts_fork_legal");
    if (on && m->d.cfg.move_algorithm != TRL_ALGO_CONVOLUTIONAL)
        return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: the tower launch runs the 'convolutional' generator only");
    if (m->legal_owed) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: a selection is waiting for trl_mcts_tower16_legal");
    m->legal_in_tower = on ? 1 : 0;
    return TRL_OK;
}

int trl_mcts_zcache_dim(trl_engine *h, int floats_per_board) {
    if (!h || !h->mcts || floats_per_board < 1 || floats_per_board > 8 * 400)
        return fail(TRL_E_INVALID, "trl_mcts_zcache_dim: %s", "bad argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    DevMcts &d = h->mcts->d;
    if (floats_per_board == d.zdim) return TRL_OK;
    HIPCHK(hipFree(d.zcache));
    d.zcache = nullptr;
    d.zdim = floats_per_board;
    HIPCHK(hipMalloc(&d.zcache, (size_t)d.n_games * d.cap_states * 2 * d.zdim * sizeof(float)));
    return TRL_OK;
}

int trl_mcts_zcache(trl_engine *h, float *dev_z) {
    if (!h || !h->mcts || !dev_z) return fail(TRL_E_INVALID, "trl_mcts_zcache: %s", "null argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_zcache, dim3(h->cfg.n_games), dim3(128), 0, h->stream, h->mcts->d, dev_z);
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

int trl_mcts_expand(trl_engine *h, const float *dev_values, const float *dev_policy) {
    if (!h || !h->mcts || !dev_values || !dev_policy) return fail(TRL_E_INVALID, "trl_mcts_expand: %s", "null argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int rcj = mcts_join(h); if (rcj != TRL_OK) return rcj; }
    if (h->mcts->legal_owed) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: the expand needs the leaf's legal moves -- trl_mcts_tower16_legal first");
    hipLaunchKernelGGL(k_expand<0>, dim3(h->cfg.n_games), dim3(BLOCK), 0, h->stream, h->mcts->d, h->d_states, dev_values,
                       dev_policy, (const float *)nullptr, (const float *)nullptr, 0, HeadsDev{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, 0});
    HIPCHK(hipGetLastError());
    h->mcts->ticks++;
    return TRL_OK;
}

int trl_mcts_expand_head(trl_engine *h, const float *dev_values, const float *dev_feats, int feat_dim,
                         const float *dev_head_weight, const float *dev_head_bias) {
    if (!h || !h->mcts || !dev_values || !dev_feats || !dev_head_weight || !dev_head_bias || feat_dim <= 0 || feat_dim > HEAD_MAX_FEATS)
        return fail(TRL_E_INVALID, "trl_mcts_expand_head: %s", "bad argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    { int rcj = mcts_join(h); if (rcj != TRL_OK) return rcj; }
    if (h->mcts->legal_owed) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: the expand needs the leaf's legal moves -- trl_mcts_tower16_legal first");
    hipLaunchKernelGGL(k_expand<1>, dim3(h->cfg.n_games), dim3(BLOCK), 0, h->stream, h->mcts->d, h->d_states, dev_values,
                       dev_feats, dev_head_weight, dev_head_bias, feat_dim, HeadsDev{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0, nullptr, 0});
    HIPCHK(hipGetLastError());
    h->mcts->ticks++;
    return TRL_OK;
}

int trl_mcts_expand_alphasame_masked(trl_engine *h, const float *dev_z, const float *dev_vec, const trl_alphasame_heads *w,
                                     const uint8_t *dev_slot_mask, int want, int last) {
    if (!h || !h->mcts || !dev_z || !dev_vec || !w || !w->oside_w || !w->oside_b || !w->value0_w || !w->value0_b ||
        !w->value3_w || !w->policy_w || !w->policy_b)
        return fail(TRL_E_INVALID, "trl_mcts_expand_alphasame: %s", "null argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->mcts->d.zdim != 400) return fail(TRL_E_STATE, "%s", "trl_mcts_expand_alphasame needs the 400-float tower cache (trl_mcts_zcache_dim)");
    HeadsDev H{w->oside_w, w->oside_b, w->value0_w, w->value0_b, w->value3_w, w->value3_b, w->use_tanh, dev_slot_mask, want};
    { int rcj = mcts_join(h); if (rcj != TRL_OK) return rcj; }
    if (h->mcts->legal_owed) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: the expand needs the leaf's legal moves -- trl_mcts_tower16_legal first");
    hipLaunchKernelGGL(k_expand<2>, dim3(h->cfg.n_games), dim3(BLOCK), 0, h->stream, h->mcts->d, h->d_states, dev_vec,
                       dev_z, w->policy_w, w->policy_b, 521, H);
    HIPCHK(hipGetLastError());
    if (last) h->mcts->ticks++;
    return TRL_OK;
}

int trl_mcts_expand_alphasame(trl_engine *h, const float *dev_z, const float *dev_vec, const trl_alphasame_heads *w) {
    return trl_mcts_expand_alphasame_masked(h, dev_z, dev_vec, w, nullptr, 0, 1);
}

// trl_mcts_expand_alphasame followed by trl_mcts_select_need of the next simulation, as one launch (k_expand_select)
int trl_mcts_expand_alphasame_select(trl_engine *h, const float *dev_z, float *dev_vec, const trl_alphasame_heads *w,
                                     float *dev_grids, uint8_t *dev_need) {
    if (!h || !h->mcts || !dev_z || !dev_vec || !w || !dev_grids)
        return fail(TRL_E_INVALID, "trl_mcts_expand_alphasame_select: %s", "null argument / no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->mcts->d.zdim != 400) return fail(TRL_E_STATE, "%s", "trl_mcts_expand_alphasame_select needs the 400-float tower cache (trl_mcts_zcache_dim)");
    if (h->mcts->fork_legal) return fail(TRL_E_STATE, "%s", "trl_mcts_expand_alphasame_select: not with trl_mcts_fork_legal");
    if (h->mcts->d.cfg.move_algorithm != TRL_ALGO_CONVOLUTIONAL)
        return fail(TRL_E_STATE, "%s", "trl_mcts_expand_alphasame_select: the fused launch runs the 'convolutional' generator only");
    HeadsDev H{w->oside_w, w->oside_b, w->value0_w, w->value0_b, w->value3_w, w->value3_b, w->use_tanh, nullptr, 0};
    if (h->mcts->legal_owed) return fail(TRL_E_STATE, "%s", "trl_mcts_legal_in_tower: the expand needs the leaf's legal moves -- trl_mcts_tower16_legal first");
    if (h->mcts->legal_in_tower && dev_need) {
        hipLaunchKernelGGL(k_expand_select<false>, dim3(h->cfg.n_games), dim3(FUSED_THREADS), 0, h->stream, h->mcts->d, h->d_states, dev_vec,
                           dev_z, w->policy_w, w->policy_b, H, dev_grids, dev_need);
        h->mcts->legal_owed = true;
    } else
        hipLaunchKernelGGL(k_expand_select<true>, dim3(h->cfg.n_games), dim3(FUSED_THREADS), 0, h->stream, h->mcts->d, h->d_states, dev_vec,
                           dev_z, w->policy_w, w->policy_b, H, dev_grids, dev_need);
    HIPCHK(hipGetLastError());
    h->mcts->ticks++;
    return TRL_OK;
}

int trl_mcts_drain_records(trl_engine *h, trl_record *host_out, int cap, int *n_out) {
    if (!h || !h->mcts || !host_out || !n_out || cap < 0) return fail(TRL_E_INVALID, "trl_mcts_drain_records: %s", "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    DevMcts &d = h->mcts->d;
    unsigned int cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, d.rec_count, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (cnt > (unsigned)d.rec_cap) cnt = (unsigned)d.rec_cap;
    if ((int)cnt > cap) return fail(TRL_E_OVERFLOW, "trl_mcts_drain_records: %s", "host buffer smaller than the pending records");
    if (cnt) HIPCHK(hipMemcpyAsync(host_out, d.recs, (size_t)cnt * sizeof(trl_record), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemsetAsync(d.rec_count, 0, 4, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    *n_out = (int)cnt;
    return TRL_OK;
}

int trl_mcts_drain_records_dev(trl_engine *h, trl_record *dev_out, int cap, int *n_out) {
    if (!h || !h->mcts || !dev_out || !n_out || cap < 0) return fail(TRL_E_INVALID, "trl_mcts_drain_records_dev: %s", "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    DevMcts &d = h->mcts->d;
    unsigned int cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, d.rec_count, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (cnt > (unsigned)d.rec_cap) cnt = (unsigned)d.rec_cap;
    if ((int)cnt > cap) return fail(TRL_E_OVERFLOW, "trl_mcts_drain_records_dev: %s", "buffer smaller than the pending records");
    if (cnt) HIPCHK(hipMemcpyAsync(dev_out, d.recs, (size_t)cnt * sizeof(trl_record), hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(hipMemsetAsync(d.rec_count, 0, 4, h->stream));
    *n_out = (int)cnt;
    return TRL_OK;
}

// ---- records out without a host round trip ---------------------------------------------------------------------------
// trl_mcts_drain_records_dev reads the pending count on the host: the stream drains, and a job ticking in chunks idles for
// the host's turnaround at every chunk.  Here the count stays on the device: one launch appends the pending records to
// the caller's buffer behind the ones already there, a second one (same stream, so after every block of the first has
// read the counts) advances the caller's count, snapshots the search counters next to it and clears the pending count.
__global__ void __launch_bounds__(256) k_drain_copy(const trl_record *src, const unsigned int *rec_count, int rec_cap,
                                                    trl_record *dst, const unsigned long long *head, int dst_cap, unsigned int slot_offset) {
    unsigned int cnt = *rec_count;
    if (cnt > (unsigned)rec_cap) cnt = (unsigned)rec_cap;
    unsigned long long base = head[0];
    unsigned long long room = base < (unsigned long long)dst_cap ? (unsigned long long)dst_cap - base : 0ull;
    if (cnt > room) cnt = (unsigned int)room;
    constexpr unsigned int W = sizeof(trl_record) / 8;               // 299 eight-byte words per record
    static_assert(sizeof(trl_record) % 8 == 0, "records are copied in 8-byte words");
    const unsigned long long *s = reinterpret_cast<const unsigned long long *>(src);
    unsigned long long *d = reinterpret_cast<unsigned long long *>(dst + base);
    size_t total = (size_t)cnt * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long v = s[i];
        if (i % W == 0) v = (v & 0xffffffff00000000ull) | (unsigned long long)((unsigned int)v + slot_offset);   // word 0 = slot | episode << 32
        d[i] = v;
    }
}

__global__ void k_drain_commit(unsigned int *rec_count, int rec_cap, unsigned long long *head, int dst_cap, unsigned long long *counters) {
    unsigned int cnt = *rec_count;
    if (cnt > (unsigned)rec_cap) cnt = (unsigned)rec_cap;
    unsigned long long base = head[0];
    unsigned long long room = base < (unsigned long long)dst_cap ? (unsigned long long)dst_cap - base : 0ull;
    unsigned int n = cnt > room ? (unsigned int)room : cnt;
    if (n < cnt) counters[3] += cnt - n;                              // records_dropped: the caller's buffer was full
    head[0] = base + n;
    for (int k = 0; k < 8; ++k) head[1 + k] = counters[k];
    *rec_count = 0;
}

int trl_mcts_drain_records_async(trl_engine *h, trl_record *dev_out, int cap, uint64_t *dev_head, uint32_t slot_offset) {
    if (!h || !h->mcts || !dev_out || !dev_head || cap < 0) return fail(TRL_E_INVALID, "trl_mcts_drain_records_async: %s", "bad argument");
    if (((uintptr_t)dev_out & 7) || ((uintptr_t)dev_head & 7)) return fail(TRL_E_INVALID, "trl_mcts_drain_records_async: %s", "buffers must be 8-byte aligned");
    HIPCHK(hipSetDevice(h->cfg.device));
    DevMcts &d = h->mcts->d;
    hipLaunchKernelGGL(k_drain_copy, dim3(256), dim3(256), 0, h->stream, d.recs, d.rec_count, d.rec_cap, dev_out,
                       (const unsigned long long *)dev_head, cap, slot_offset);
    hipLaunchKernelGGL(k_drain_commit, dim3(1), dim3(1), 0, h->stream, d.rec_count, d.rec_cap, (unsigned long long *)dev_head, cap, d.counters);
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

int trl_mcts_get_stats(trl_engine *h, trl_mcts_stats *out) {
    if (!h || !h->mcts || !out) return fail(TRL_E_INVALID, "trl_mcts_get_stats: %s", "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    unsigned long long c[8];
    HIPCHK(hipMemcpyAsync(c, h->mcts->d.counters, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out->ticks = c[6]; out->simulations = c[0]; out->moves = c[1]; out->games = c[2];
    out->records_dropped = c[3]; out->max_nodes_used = (uint32_t)c[4]; out->overflow = (uint32_t)c[5];
    out->boards_reused = c[7];
    return TRL_OK;
}

int trl_mcts_leaf_states(trl_engine *h, trl_game_state *host_out, int32_t *host_info) {
    if (!h || !h->mcts || !host_out || !host_info) return fail(TRL_E_INVALID, "trl_mcts_leaf_states: %s", "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    int n = h->cfg.n_games;
    { int rcj = mcts_join(h); if (rcj != TRL_OK) return rcj; }
    hipLaunchKernelGGL(k_leaf_gather, dim3(n), dim3(128), 0, h->stream, h->mcts->d, h->mcts->d_leaf_tmp, h->mcts->d_info_tmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(host_out, h->mcts->d_leaf_tmp, (size_t)n * sizeof(trl_game_state), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(host_info, h->mcts->d_info_tmp, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return TRL_OK;
}

int trl_mcts_root_children(trl_engine *h, int32_t *counts, uint16_t *moves, int32_t *visits, double *q, double *p) {
    if (!h || !h->mcts || !counts || !moves || !visits || !q || !p) return fail(TRL_E_INVALID, "trl_mcts_root_children: %s", "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    size_t n = (size_t)h->cfg.n_games, L = TRL_MAX_LEGAL;
    char *base = (char *)h->mcts->d_root_tmp;
    double *dq = (double *)base, *dp = dq + n * L;
    int32_t *dv = (int32_t *)(dp + n * L), *dc = dv + n * L;
    uint16_t *dm = (uint16_t *)(dc + n);
    hipLaunchKernelGGL(k_root_children, dim3((unsigned)n), dim3(128), 0, h->stream, h->mcts->d, dc, dm, dv, dq, dp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(counts, dc, n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(moves, dm, n * L * 2, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(visits, dv, n * L * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(q, dq, n * L * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(p, dp, n * L * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return TRL_OK;
}

// test hook: the stub evaluator on the current leaves (values[n], policy[n][11583], device pointers)
int trl_debug_stub_eval_salted(trl_engine *h, float *dev_values, float *dev_policy, uint64_t salt);
#ifdef TRL_PROFILE
int trl_debug_mcts_slotprof(unsigned int *host_out, int n_slots) {
    HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_slotprof), sizeof(unsigned int) * 16 * (size_t)n_slots));
    return TRL_OK;
}
int trl_debug_mcts_prof(unsigned long long *host_out, int reset) {
    HIPCHK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_mprof), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_mprof), z, sizeof z)); }
    return TRL_OK;
}
#endif

int trl_debug_stub_eval(trl_engine *h, float *dev_values, float *dev_policy) {
    return trl_debug_stub_eval_salted(h, dev_values, dev_policy, 0);
}
// salt != 0 gives a different deterministic evaluator ("a second network") for the gating tests
int trl_debug_stub_eval_salted(trl_engine *h, float *dev_values, float *dev_policy, uint64_t salt) {
    if (!h || !h->mcts) return fail(TRL_E_INVALID, "%s", "no search attached");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_stub_eval, dim3(h->cfg.n_games), dim3(256), 0, h->stream, h->mcts->d, dev_values, dev_policy, salt);
    HIPCHK(hipGetLastError());
    return TRL_OK;
}

}  // extern "C"

// trl_net.hip asks for the job when it launches the tower: true (and the job filled in) when a selection of this engine
// left its legal-move search to the tower launch
bool trl_mcts_leaf_legal_job(trl_engine *h, LeafLegalJob *job) {
    if (!h || !h->mcts || !h->mcts->legal_in_tower) return false;
    DevMcts &d = h->mcts->d;
    *job = LeafLegalJob{d.ctl, d.pool, d.leaf_moves, d.counters, d.cap_states, d.n_games};
    h->mcts->legal_owed = false;
    return true;
}

