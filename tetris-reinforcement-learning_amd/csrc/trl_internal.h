// trl_internal.h -- pieces shared by the translation units of libtrl_hip.so (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/tetris_engine.h"
#include "trl_device.h"

namespace trl {


constexpr int BLOCK = 128;
constexpr int STATE_DW = sizeof(trl_game_state) / 4;    // 80

struct BlockShared {
    alignas(16) trl_game_state g;
    MgScratch mg[2];
    alignas(4) uint16_t list[2][TRL_MAX_LEGAL];
    int cnt[2];
    int order;          // which slot's list comes first in plane order
    int overflow;
    int chosen;         // rollout: encoded move picked for this ply (or -1)
    int done;
};

__device__ __forceinline__ void load_state(BlockShared &sh, const trl_game_state *src, int tid) {
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
    uint32_t *d = reinterpret_cast<uint32_t *>(&sh.g);
    if (tid < STATE_DW) d[tid] = s[tid];
}
__device__ __forceinline__ void store_state(const BlockShared &sh, trl_game_state *dst, int tid) {
    const uint32_t *s = reinterpret_cast<const uint32_t *>(&sh.g);
    uint32_t *d = reinterpret_cast<uint32_t *>(dst);
    if (tid < STATE_DW) d[tid] = s[tid];
}

// legal placements of the side to move into sh.list / sh.cnt; all 128 threads call it.
// Returns (uniformly) the total count; list order = slot sh.order first.
static __device__ __forceinline__ int block_movegen(BlockShared &sh, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) { sh.overflow = 0; sh.cnt[0] = 0; sh.cnt[1] = 0; sh.order = 0; }
    __syncthreads();
    const trl_game_state &g = sh.g;
    bool live = !is_terminal(g) && !no_move(g);
    if (live) {
        const trl_player_state &p = g.p[g.turn];
        int t0, t1;
        movegen_types(p, &t0, &t1);
        int type = wave == 0 ? t0 : t1;
        if (type != P_NONE) {
            int ovf = 0;
            int n = movegen_wave(p.rows, type, wave == 1, sh.mg[wave], sh.list[wave], TRL_MAX_LEGAL, lane, &ovf);
            if (lane == 0) { sh.cnt[wave] = n; if (ovf) sh.overflow = 1; }
        }
        if (tid == 0) {
            // planes of different piece types are disjoint ranges: lower base first (argwhere order)
            int b0 = t0 != P_NONE ? C_PLANE_BASE[t0] : 99, b1 = t1 != P_NONE ? C_PLANE_BASE[t1] : 99;
            sh.order = b1 < b0 ? 1 : 0;
        }
    }
    __syncthreads();
    int total = sh.cnt[0] + sh.cnt[1];
    if (total > TRL_MAX_LEGAL) { total = TRL_MAX_LEGAL; if (tid == 0) sh.overflow = 1; }
    return total;
}
__device__ __forceinline__ uint16_t merged_move(const BlockShared &sh, int i) {
    int a = sh.order, na = sh.cnt[a];
    return i < na ? sh.list[a][i] : sh.list[1 - a][i - na];
}

// The same for a workgroup of 256 threads whose waves 2 and 3 would otherwise idle (the legal-move job of the tower launch,
// trl_net.hip): waves 0 and 1 search the two piece slots with the closure and the mixed-cell rule; beside them waves 2 and
// 3 start the ORDERED WALK of a slot whose piece is T -- speculatively: 96 % of the T searches do not need it and call it
// off (`stop`, looked at once per batch of queue items), but for the ones that do the walk is then well on its way instead
// of starting when the closure gives up -- closure 8 us + walk 32 us one after the other was what a launch with 256 slots
// lasted in four launches of five (48 us against the tower's 40).  Same lists as block_movegen.
struct SpecShared {
    MgScratch mg[2];
    alignas(4) uint16_t list[2][TRL_MAX_LEGAL];
    int cnt[2], ovf[2];
    int stop[2];            // set by the closure wave of the slot: the walk's result is not needed
    int need[2];            // set by the closure wave of the slot: it is
};
static __device__ __forceinline__ int block_movegen_spec(BlockShared &sh, SpecShared &sp, int tid) {
    const int wave = tid >> 6, lane = tid & 63, slot = wave & 1;
    if (tid == 0) {
        sh.overflow = 0; sh.cnt[0] = 0; sh.cnt[1] = 0; sh.order = 0;
        sp.stop[0] = 0; sp.stop[1] = 0; sp.need[0] = 0; sp.need[1] = 0; sp.cnt[0] = 0; sp.cnt[1] = 0; sp.ovf[0] = 0; sp.ovf[1] = 0;
    }
    __syncthreads();
    const trl_game_state &g = sh.g;
    const bool live = !is_terminal(g) && !no_move(g);
    if (live) {
        const trl_player_state &p = g.p[g.turn];
        int t0, t1;
        movegen_types(p, &t0, &t1);
        const int type = slot == 0 ? t0 : t1;
        if (wave < 2) {
            if (type != P_NONE) {
                int ovf = 0;
                CsScratch &C = reinterpret_cast<CsScratch &>(sh.mg[slot]);
                int n = movegen_closure_wave(p.rows, type, slot == 1, C, sh.list[slot], TRL_MAX_LEGAL, lane, &ovf);
                if (n == -2) n = movegen_closure_finish_mixed(C, sh.list[slot], TRL_MAX_LEGAL, lane, &ovf);
                if (lane == 0) {
                    if (n >= 0) { sh.cnt[slot] = n; if (ovf) sh.overflow = 1; sp.stop[slot] = 1; }
                    else sp.need[slot] = 1;
                }
            }
            if (tid == 0) {
                int b0 = t0 != P_NONE ? C_PLANE_BASE[t0] : 99, b1 = t1 != P_NONE ? C_PLANE_BASE[t1] : 99;
                sh.order = b1 < b0 ? 1 : 0;
            }
        } else if (type == PT) {
            int ovf = 0;
            const int n = movegen_ordered_wave(p.rows, type, slot == 1, sp.mg[slot], sp.list[slot], TRL_MAX_LEGAL, lane, &ovf, &sp.stop[slot]);
            if (lane == 0) { sp.cnt[slot] = n; sp.ovf[slot] = ovf; }
        }
    }
    __syncthreads();
    if (live && wave < 2 && sp.need[slot]) {              // the closure stopped at cells the rule does not settle: the walk's list
        const int n = sp.cnt[slot];
        for (int i = lane; i < n; i += 64) sh.list[slot][i] = sp.list[slot][i];
        if (lane == 0) { sh.cnt[slot] = n; if (sp.ovf[slot]) sh.overflow = 1; }
    }
    __syncthreads();
    int total = sh.cnt[0] + sh.cnt[1];
    if (total > TRL_MAX_LEGAL) { total = TRL_MAX_LEGAL; if (tid == 0) sh.overflow = 1; }
    return total;
}

// the same with the reference's default generator, 'brute-force' (move_generation.py:204-243; row M6): k_legal_bruteforce
// and the search when Config.move_algorithm says so (ai.py:84, 414).  48 KB of LDS per workgroup (the cell-by-cell queue).
struct BfBlockShared {
    alignas(16) trl_game_state g;
    BfScratch mg[2];
    alignas(4) uint16_t list[2][TRL_MAX_LEGAL];
    int cnt[2], order, overflow;
};
static __device__ __forceinline__ int block_movegen(BfBlockShared &sh, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) { sh.overflow = 0; sh.cnt[0] = 0; sh.cnt[1] = 0; sh.order = 0; }
    __syncthreads();
    const trl_game_state &g = sh.g;
    if (!is_terminal(g) && !no_move(g)) {
        const trl_player_state &p = g.p[g.turn];
        int t0, t1;
        movegen_types(p, &t0, &t1);
        const int type = wave == 0 ? t0 : t1;
        if (type != P_NONE) {
            int ovf = 0;
            int n = movegen_bruteforce_wave(p.rows, type, wave == 1, sh.mg[wave], sh.list[wave], TRL_MAX_LEGAL, lane, &ovf);
            if (lane == 0) { sh.cnt[wave] = n; if (ovf) sh.overflow = 1; }
        }
        if (tid == 0) {
            int b0 = t0 != P_NONE ? C_PLANE_BASE[t0] : 99, b1 = t1 != P_NONE ? C_PLANE_BASE[t1] : 99;
            sh.order = b1 < b0 ? 1 : 0;
        }
    }
    __syncthreads();
    int total = sh.cnt[0] + sh.cnt[1];
    if (total > TRL_MAX_LEGAL) { total = TRL_MAX_LEGAL; if (tid == 0) sh.overflow = 1; }
    return total;
}
__device__ __forceinline__ uint16_t merged_move(const BfBlockShared &sh, int i) {
    int a = sh.order, na = sh.cnt[a];
    return i < na ? sh.list[a][i] : sh.list[1 - a][i - na];
}


// ---- the search's per-slot control block (trl_mcts.hip), and the part of a tick that may run elsewhere -------------------
struct SlotCtl {
    int32_t phase;          // 0 = start a new search at the next select, 1 = searching, 2 = idle (game over, no auto-reset)
    int32_t iter, max_iter, fast;
    int32_t n_nodes, n_states;
    int32_t leaf, leaf_kind, leaf_nlegal, leaf_state;
    int32_t overflow, leaf_parent_state;      // state index of the leaf's parent (-1 at the root)
    // what the selection knows and the expand would otherwise re-load through dependent global reads: the leaf's side to
    // move, and its parent's node index and child range (FPU refresh of the siblings, ai.py:536-565)
    int32_t leaf_turn, par_node, par_first, par_n;
    int32_t leaf_zsrc, reserved_;             // >= 0: state (| player << 30) whose cached tower output is the mover's board's (DevMcts::btab)
    int32_t depth, random_move;               // nodes on the selection path root .. leaf (DevMcts::path); this search picks a
    double  terminal_value;                   // random starting move (ai.py:1588-1608)
    double  rand_left;                        // num_random_moves still to play in this game
};

// The legal-move search of every slot's current leaf (get_move_matrix + get_move_list, ai.py:414-417) needs the leaf's
// position only -- like the net's tower, and unlike everything after it in the tick.  Split off the selection it runs
// either as k_leaf_legal on a second stream (trl_mcts_fork_legal) or as extra workgroups of the tower launch
// (trl_mcts_tower16_legal, trl_net.hip): 64 VGPRs and 10 KB of LDS per workgroup, in the registers the tower's boards leave
// free, and a T search that falls back to the ordered walk (3 % of the searches, 4 x the time) no longer sets the length
// of the search kernel's launch.
struct LeafLegalJob {
    SlotCtl *ctl;                       // [n_slots]: phase, leaf_kind, leaf_state read; leaf_nlegal, overflow written
    const trl_game_state *pool;         // [n_slots][cap_states]
    uint16_t *leaf_moves;               // [n_slots][TRL_MAX_LEGAL]
    unsigned long long *counters;       // [5] += 1 when a legal-move cap was hit
    int cap_states, n_slots;
};

// all 256 threads of a workgroup call it (block_movegen_spec: the ordered walk of a T search starts beside the closure)
static __device__ __forceinline__ void leaf_legal_block_spec(const LeafLegalJob &J, BlockShared &sh, SpecShared &sp, int slot, int tid) {
    const int phase = J.ctl[slot].phase, kind = J.ctl[slot].leaf_kind, sidx = J.ctl[slot].leaf_state;
    if (phase != 1 || kind != 0) return;
    load_state(sh, J.pool + (size_t)slot * J.cap_states + sidx, tid);
    __syncthreads();
    const int nlegal = block_movegen_spec(sh, sp, tid);
    for (int i = tid; i < nlegal; i += 2 * BLOCK) J.leaf_moves[(size_t)slot * TRL_MAX_LEGAL + i] = merged_move(sh, i);
    if (tid == 0) {
        J.ctl[slot].leaf_nlegal = nlegal;
        if (sh.overflow) { J.ctl[slot].overflow = 1; atomicAdd(&J.counters[5], 1ull); }
    }
}

// all BLOCK (128) threads of a workgroup call it
static __device__ __forceinline__ void leaf_legal_block(const LeafLegalJob &J, BlockShared &sh, int slot, int tid) {
    const int phase = J.ctl[slot].phase, kind = J.ctl[slot].leaf_kind, sidx = J.ctl[slot].leaf_state;
    if (phase != 1 || kind != 0) return;
    load_state(sh, J.pool + (size_t)slot * J.cap_states + sidx, tid);
    __syncthreads();
    const int nlegal = block_movegen(sh, tid);
    for (int i = tid; i < nlegal; i += BLOCK) J.leaf_moves[(size_t)slot * TRL_MAX_LEGAL + i] = merged_move(sh, i);
    if (tid == 0) {
        J.ctl[slot].leaf_nlegal = nlegal;
        if (sh.overflow) { J.ctl[slot].overflow = 1; atomicAdd(&J.counters[5], 1ull); }
    }
}

}  // namespace trl

struct trl_mcts;   // search state (trl_mcts.hip)
struct trl_engine;
void trl_mcts_free_internal(trl_engine *h);   // trl_mcts.hip
bool trl_mcts_leaf_legal_job(trl_engine *h, trl::LeafLegalJob *job);   // trl_mcts.hip: false unless trl_mcts_legal_in_tower is on

struct trl_engine {
    trl_config cfg;
    hipStream_t stream;
    bool own_stream;
    trl_game_state *d_states;
    uint16_t *d_counts, *d_moves, *d_in_moves;
    unsigned long long *d_steps;   // two blocks of [step count, 16 bytes][plies per slot]: a launch uses one and zeroes the other
    unsigned int *d_prog;
    size_t pp_stride;              // bytes per block
    int pp;                        // block of the next launch
    unsigned long long *h_steps;   // pinned host word the step count is copied to
    hipEvent_t ev0, ev1;
    bool have_reset;
    float last_ms;
    trl_mcts *mcts;
    int n_cu = 0;                          // compute units of the device (rollout pacing: are all slots resident?)
    hipGraph_t graph = nullptr;            // captured search tick (trl_graph_begin / _end / _launch)
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph_many = nullptr;       // TRL_GRAPH_UNROLL copies of it in a chain: one launch, no gap between the ticks
    hipGraphExec_t graph_many_exec = nullptr;
    int graph_unroll = 1;
};

extern thread_local char trl_g_err[512];
inline int fail(int code, const char *fmt, const char *detail) {
    snprintf(trl_g_err, sizeof trl_g_err, fmt, detail);
    return code;
}
#define HIPCHK(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) { (void)hipGetLastError(); /* reported here: do not leave it for the next HIP user */ \
                                return fail(TRL_E_HIP, #call ": %s", hipGetErrorString(e_)); } \
    } while (0)

