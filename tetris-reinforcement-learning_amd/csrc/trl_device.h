// trl_device.h -- device functions of the env-step path (gfx950, wave64).
//
// Mapping to the hardware
//   * one wavefront per (game, piece type) runs the legal-placement search of
//     move_generation.py:325-488 on 14-bit row masks held in LDS: the search itself is an
//     ordered FIFO walk (the T-spin sub-plane of a placement depends on the order in which
//     kicks reach it, move_generation.py:671-677), so the FIFO is drained in order while the
//     work inside one item is spread over the lanes -- queue items are classified 64 at a
//     time, a flooded row's (edge column x rotation direction) kick probes run one per lane,
//     and pushes are appended with ballot + popcount so the queue keeps the reference order;
//   * the board is staged once per step in LDS as 40 ten-bit rows;
//   * one lane applies the chosen placement (lock, line clear, attack, garbage, spawn).
//
// Every function cites the reference lines it re-implements; the arithmetic is integer-only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tetris_engine.h"
#include "trl_tables.h"

namespace trl {

constexpr int MAP_W = 14;          // validity-map width  (COLS + 4), x in [-2, 11]
constexpr int MAP_H = 44;          // validity-map height (ROWS + 4), y in [-2, 41]
constexpr int QCAP  = 1024;        // FIFO capacity per search (observed max 511 in random play)
constexpr int SPAWN_Y = TRL_ROWS - TRL_SPAWN_ROW;   // 17

// LDS scratch of one wavefront's search
// Member order matters: rows 0, 1, 42 and 43 of every validity map are all-zero (a piece origin that high or low
// puts a cell outside the 40 rows) and visited / present bits are subsets of valid ones, so the two rows before and
// after valid[r] -- the end of visited[3] or valid[r-1], the start of valid[r+1] or present[0] -- are zero guard rows:
// the kick probes read valid[r][y] for y in [-2, 45] without clamping or bounds tests.
struct alignas(8) MgScratch {
    uint16_t visited[4][MAP_H];
    uint16_t valid[4][MAP_H];      // valid[r][ym] bit xm: piece (rotation r) at origin (xm-2, ym-2) fits; bits 14, 15 zero
    uint16_t present[4][MAP_H];    // unflagged placements
    uint16_t flagp[4][MAP_H];      // placements reached by a rotation (rotation_just_occurred)
    uint16_t ulp[4][MAP_H];        // ... whose last writer used the last kick of its list
    uint16_t queue[QCAP];          // mx:4 | my:6 | rot:2 | flag:1 | last_kick:1
    uint16_t edge[MAP_H];          // edge-cell mask of each row of the flood in progress
};

// ---- wave helpers -------------------------------------------------------------------------
__device__ __forceinline__ void wave_sync() {
    // LDS accesses of one wavefront execute in order; this only pins the compiler.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int lane_rank(uint64_t mask, int lane) {
    return __popcll(mask & ((1ull << lane) - 1ull));
}
// inclusive prefix sum over the 64 lanes (all active) with DPP moves in the vector ALU: four row_shr steps scan each
// row of 16 lanes, row_bcast:15 / row_bcast:31 carry the row totals across (rows 1, 3 and rows 2, 3).  A scan
// through __shfl_up is six dependent ds_bpermute round trips instead.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// ---- rng (include/tetris_engine.h) ----------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t rng_below(uint64_t seed, uint32_t stream, uint32_t purpose, uint64_t ctr, uint32_t n) {
    uint64_t k = mix64(seed ^ ((uint64_t)stream * 0xD1342543DE82EF95ull));
    uint32_t u = (uint32_t)(mix64(k ^ ((uint64_t)purpose << 56) ^ ctr) >> 32);
    return (uint32_t)(((uint64_t)u * n) >> 32);
}

// ---- board primitives on bit rows -----------------------------------------------------------
// Board.is_valid_position over the 4 minos of a pose (board.py:23-30, piece.py:50-55)
__device__ __forceinline__ bool pose_fits(const uint16_t *rows, int type, int x, int y, int rot) {
    uint32_t m = C_MINO[type * 4 + rot];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int col = x + (int)((m >> (4 * i)) & 3), row = y + (int)((m >> (4 * i + 2)) & 3);
        bool in = (unsigned)row < (unsigned)TRL_ROWS && (unsigned)col < (unsigned)TRL_COLS;
        ok = ok && in && !((rows[in ? row : 0] >> (in ? col : 0)) & 1);
    }
    return ok;
}
__device__ __forceinline__ bool cell_free(const uint16_t *rows, int col, int row) {
    bool in = (unsigned)row < (unsigned)TRL_ROWS && (unsigned)col < (unsigned)TRL_COLS;
    return in && !((rows[in ? row : 0] >> (in ? col : 0)) & 1);
}
__device__ __forceinline__ int spawn_x(int type) { return 3 + (type == PO ? 1 : 0); }   // piece.py:17-21

// one row of the validity map (move_generation.py:490-528): AND over the minos of the
// shifted free-cell rows
__device__ __forceinline__ uint32_t valid_row(const uint16_t *rows, int type, int rot, int ym) {
    uint32_t m = C_MINO[type * 4 + rot];
    uint32_t v = 0x3FFFu;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int cx = (int)((m >> (4 * i)) & 3), cy = (int)((m >> (4 * i + 2)) & 3);
        int r = ym - 2 + cy;
        uint32_t fr = ((unsigned)r < (unsigned)TRL_ROWS) ? ((~(uint32_t)rows[r]) & 0x3FFu) : 0u;
        v &= (fr << 2) >> cx;
    }
    return v & 0x3FFFu;
}

// all cells of `seed` plus every cell connected to one through left/right steps inside `m`
// (the while-loop of move_generation.py:418-422 as two carry chains)
__device__ __forceinline__ uint32_t row_fill(uint32_t m, uint32_t seed) {
    uint32_t up = (((m + seed) ^ m) & m) | seed;
    uint32_t mr = __brev(m) >> 16, sr = __brev(seed) >> 16;
    uint32_t dn = (((mr + sr) ^ mr) & mr) | sr;
    return up | (__brev(dn) >> 16);
}

// ---- flagged placements, kept in the reference's "last flagged writer wins" order ------------
// (move_generation.py:671-677: unique_placements[key] = loc whenever loc is flagged).  The emitting lanes of one
// call are in the reference's order; per key only the LAST of them decides the last-kick bit.  "Am I the last
// emitter of my key" is settled in registers (one readlane per emitter), then every deciding lane updates the
// planes at once with LDS atomics on the 32-bit words holding the 16-bit row masks.
__device__ __forceinline__ void emit_flagged(MgScratch &S, bool emit, int rot, int my, uint32_t bit, int ul, int lane, bool isT) {
    // Only the T planes look at the flags.  For the other pieces a flagged placement is one more way to reach a cell
    // that is stuck and visited -- and every visited stuck cell got its `present` bit from the flood that visited it
    // (blocked_down), so the flag adds nothing to present | flagged (move_generation.py:650-749 merges them).
    if (!isT) return;
    uint64_t m = __ballot(emit);
    if (!m) return;
    const int idx = rot * MAP_H + my;
    const uint32_t wbit = bit << ((idx & 1) * 16);
    unsigned int *fw = reinterpret_cast<unsigned int *>(&S.flagp[0][0]) + (idx >> 1);
    {
        const int key = (idx << 4) | (31 - __clz((int)bit));
        bool last = emit;
        for (uint64_t mm = m; mm; mm &= mm - 1) {
            const int l = __ffsll((unsigned long long)mm) - 1;
            const int kl = __builtin_amdgcn_readlane(key, l);
            if (key == kl && lane < l) last = false;
        }
        if (last) {
            unsigned int *uw = reinterpret_cast<unsigned int *>(&S.ulp[0][0]) + (idx >> 1);
            atomicOr(fw, wbit);
            if (ul) atomicOr(uw, wbit); else atomicAnd(uw, ~wbit);
        }
    }
    wave_sync();
}

#ifdef TRL_PROFILE
static __device__ unsigned long long g_prof[16];
static __device__ unsigned long long g_cprof[16];   // closure search: 0 set-up, 1 fixed point, 2 kicks, 3 planes + list, 4 mixed stage, 8 searches, 9 fixed-point steps, 10 kick rounds, 11 kick probes
#define PROF_T(v) unsigned long long v = __builtin_readcyclecounter()
#define PROF_ADD(i, a, b) do { prof_acc[i] += (b) - (a); } while (0)
#else
#define PROF_T(v)
#define PROF_ADD(i, a, b)
#endif

// placements -> policy planes -> ascending move list (_convert_placements_to_policy, move_generation.py:650-749;
// get_move_list order, ai.py:1016-1024).  S.present / S.flagp / S.ulp: per (rotation, row) the columns with an unflagged
// placement / a rotation-flagged one / whose last flagged writer used the last kick.  One wavefront, lane = policy row.
template <class SCR>
static __device__ __forceinline__ int placements_to_moves(const SCR &S, int type, uint16_t *out, int out_cap, int lane) {
    const bool isT = (type == PT);
    int total = 0;
    const int nrotp = C_NROT[type], base = C_PLANE_BASE[type];
    const bool zsi = (type == PZ || type == PS || type == PI);
    const int nts = isT ? 3 : 1;
    for (int ts = 0; ts < nts; ts++) {
        for (int ri = 0; ri < nrotp; ri++) {
            uint32_t m = 0;
            int row = lane;
            if (row < TRL_POLICY_ROWS) {
                auto src = [&](int o, int my) -> uint32_t {
                    uint32_t p = S.present[o][my];
                    if (!isT) return p;                                  // flagged placements are a subset (emit_flagged)
                    uint32_t f = S.flagp[o][my], u = S.ulp[o][my];
                    return ts == 0 ? (p & ~f) : (ts == 1 ? (f & ~u) : (f & u));
                };
                m = src(ri, row + 2);
                if (zsi) {
                    if (ri == 0) { if (row >= 1) m |= src(2, row + 1); }           // rot 2 -> rot 0, row + 1
                    else m |= src(3, row + 2) >> 1;                                  // rot 3 -> rot 1, col - 1
                }
                m &= 0x7FFu;
            }
            int cnt = __popc(m);
            int incl = wave_incl_scan(cnt, lane);
            int off = total + incl - cnt;
            int plane = base + ri + 4 * ts;
            while (m) {
                int c = __ffs((int)m) - 1; m &= m - 1;
                if (off < out_cap) out[off] = (uint16_t)TRL_MOVE(plane, row, c);
                off++;
            }
            total += __builtin_amdgcn_readlane(incl, 63);
        }
    }
    return total;
}

// ---- the search: _generate_moves_for_piece + _convolutional_algorithm +
//      _convert_placements_to_policy (move_generation.py:107-153, 325-488, 650-749) -----------
// One wavefront.  rows: the player's 40 board rows in LDS.  `need_spawn` = the piece comes out of
// hold / next (Player.create_piece must succeed, player.py:37-44).  Writes the encoded
// placements of this piece type in ascending order to out[] and returns their number;
// *overflow is set when a cap was hit.
// `stop`: optional LDS word another wave sets when this walk's result is no longer wanted (block_movegen_spec); looked at
// once per batch of queue items; the walk then returns -1
static __device__ __forceinline__ int movegen_ordered_wave(const uint16_t *rows, int type, bool need_spawn, MgScratch &S,
                            uint16_t *out, int out_cap, int lane, int *overflow, const int *stop = nullptr) {
#ifdef TRL_PROFILE
    unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PROF_T(p0);
    // validity maps + clear planes
    for (int i = lane; i < 4 * MAP_H; i += 64) {
        int r = i / MAP_H, ym = i - r * MAP_H;
        S.valid[r][ym] = (uint16_t)valid_row(rows, type, r, ym);
        S.visited[r][ym] = 0; S.present[r][ym] = 0;
        if (type == PT) { S.flagp[r][ym] = 0; S.ulp[r][ym] = 0; }                // only T reads the flag planes
    }
    // _get_highest_row / _set_starting_position (move_generation.py:164-180)
    uint64_t occ = __ballot(lane < TRL_ROWS && rows[lane < TRL_ROWS ? lane : 0] != 0);
    int highest = occ ? (__ffsll((unsigned long long)occ) - 1) : TRL_ROWS;
    int box = C_BOX[type];
    int sy = highest - box; if (sy < SPAWN_Y) sy = SPAWN_Y;
    int sx = spawn_x(type);
    wave_sync();

    PROF_T(p1); PROF_ADD(0, p0, p1);
    bool ok = (S.valid[0][sy + 2] >> (sx + 2)) & 1;                                  // :349-351
    if (need_spawn) ok = ok && ((S.valid[0][SPAWN_Y + 2] >> (sx + 2)) & 1);           // player.py:40
    ok = uni(ok);
    int total = 0;
    if (ok) {
        const bool check_rot = (type != PO);
        const bool isT = (type == PT);
        const int ktab = (type == PI) ? 12 : 0;
        uint16_t *task = out;      // the move-list buffer doubles as the edge-cell list of the flood in progress (my:6 | mx:4)
        int head = 0, tail = 1;
        if (lane == 0) S.queue[0] = (uint16_t)((sx + 2) | ((sy + 2) << 4));
        wave_sync();

        bool stopped = false;
        while (head < tail) {
            if (stop && uni(*reinterpret_cast<const volatile int *>(stop))) { stopped = true; break; }
            PROF_T(q0);
            // classify up to 64 queued items at once
            int idx = head + lane;
            bool act = idx < tail;
            uint32_t it = S.queue[act ? idx : head];
            int mx = it & 15, my = (it >> 4) & 63, rot = (it >> 10) & 3, fl = (it >> 12) & 1, ul = (it >> 13) & 1;
            uint32_t bit = 1u << mx;
            const uint32_t cv = S.visited[rot][my];
            bool vis = act & ((cv & bit) != 0);
            uint64_t unv = __ballot(act && !vis);
            int first = unv ? (__ffsll((unsigned long long)unv) - 1) : 64;
            // items up to and including the first unvisited one are popped now (:382-398); only T keeps their flags
            if (type == PT) {
                const uint32_t cb = S.valid[rot][my + 1 < MAP_H ? my + 1 : my];
                const bool stuck = (my + 1 >= MAP_H) | ((cb & bit) == 0);
                emit_flagged(S, act & (lane <= first) & (fl != 0) & stuck, rot, my, bit, ul, lane, true);
            }
            PROF_T(q1); PROF_ADD(1, q0, q1); PROF_ADD(8, 0ull, 1ull);
            if (first == 64) { int n = tail - head; head += (n < 64 ? n : 64); continue; }
            head += first + 1;

            const uint32_t fit = (uint32_t)__builtin_amdgcn_readlane((int)it, first);      // `first` is wave-uniform
            const int fmx = fit & 15, fmy = (fit >> 4) & 63, frot = (fit >> 10) & 3;
            if (!((S.valid[frot][fmy] >> fmx) & 1)) continue;                          // :406-407

            // (1) scanline flood fill of this rotation state, one row per iteration (:409-425, :485-488); wave-uniform.
            //     The kick probes of the reference's inner loop only read the OTHER rotations' visited maps, which
            //     this item does not change, so they are deferred until the whole flood is known.
            uint32_t r = 1u << fmx;
            int nrows = 0;
            uint32_t vr = S.valid[frot][fmy], rvis = S.visited[frot][fmy];      // the row below becomes the next row:
            for (int fy = fmy; fy < MAP_H && r; ++fy, ++nrows) {                    // two LDS reads per row, not four
                r = row_fill(vr & ~rvis, r);
                const uint32_t next_vr = (fy + 1 < MAP_H) ? (uint32_t)S.valid[frot][fy + 1] : 0u;
                const uint32_t next_vis = (fy + 1 < MAP_H) ? (uint32_t)S.visited[frot][fy + 1] : 0u;
                const uint32_t blocked_down = r & ~next_vr;
                const uint32_t edges = blocked_down | (r & ~(vr << 1)) | (r & ~(vr >> 1));
                if (lane == 0) {
                    S.visited[frot][fy] = (uint16_t)(rvis | r);
                    S.present[frot][fy] |= (uint16_t)blocked_down;                     // :439-440
                    S.edge[nrows] = (uint16_t)edges;
                }
                r = r & next_vr & ~next_vis;
                vr = next_vr; rvis = next_vis;
            }
            wave_sync();
            PROF_T(q2); PROF_ADD(2, q1, q2); PROF_ADD(9, 0ull, (unsigned long long)nrows);
            if (!check_rot) continue;
            // (2) edge cells of all rows, compacted in (row, column) order: lane = row
            int n_edges;
            {
                uint32_t e = (lane < nrows) ? (uint32_t)S.edge[lane] : 0u;
                int cnt = __popc(e);
                int incl = wave_incl_scan(cnt, lane);
                int off = incl - cnt;
                n_edges = __builtin_amdgcn_readlane(incl, 63);
                if (n_edges > out_cap) { *overflow = 1; n_edges = out_cap; }
                while (e) {
                    int c = __ffs((int)e) - 1; e &= e - 1;
                    if (off < out_cap) task[off] = (uint16_t)(((fmy + lane) << 4) | c);
                    off++;
                }
            }
            wave_sync();
            PROF_T(q3); PROF_ADD(3, q2, q3);
            // (3) kick probes (:443-481), one (edge cell, direction) per lane, 64 per pass, in the reference's order
            const int n_tasks = 3 * n_edges;
            // the three kick lists of this rotation: wave-uniform (scalar) loads, selected per lane below
            const uint64_t kt1 = C_KICK[ktab + frot * 3 + 0], kt2 = C_KICK[ktab + frot * 3 + 1], kt3 = C_KICK[ktab + frot * 3 + 2];
            for (int t0 = 0; t0 < n_tasks; t0 += 64) {
                const int t = t0 + lane;
                const bool probe = t < n_tasks;
                const int ei = t / 3, dir = t - ei * 3 + 1;
                const uint32_t ed = task[probe ? ei : 0];                // unconditional LDS reads throughout: one
                const int col = ed & 15, fy = ed >> 4;                     // round trip per stage instead of one per probe
                const int nrot = (frot + dir) & 3;
                const uint64_t kt = dir == 1 ? kt1 : (dir == 2 ? kt2 : kt3);
                const int nk = (int)(kt >> 60);
                const uint16_t *vrow = S.valid[nrot];
                // fit mask of the (up to six) kicks of this probe, bit ki = kick ki lands on a valid cell.  No bounds
                // tests: x in [-2, 15] selects a bit of the 16-bit row whose bits 14, 15 are zero, y in [-2, 45] reads
                // zero guard rows (MgScratch).  The first fitting kick is the reference's (:452-470).
                const int colm2 = col - 2, fyp2 = fy + 2;
                uint32_t fitm = 0;
#pragma unroll
                for (int ki = 0; ki < 6; ki++) {
                    const int tx = colm2 + (int)((kt >> (6 * ki)) & 7);
                    const int ty = fyp2 - (int)((kt >> (6 * ki + 3)) & 7);
                    fitm |= (((uint32_t)vrow[ty] >> (tx & 15)) & 1u) << ki;
                }
                fitm &= (1u << nk) - 1u;
                fitm = probe ? fitm : 0u;
                const int kidx = fitm ? (__ffs((int)fitm) - 1) : 0;
                const uint32_t kk2 = (uint32_t)(kt >> (6 * kidx));
                const int nmx = colm2 + (int)(kk2 & 7), nmy = fyp2 - (int)((kk2 >> 3) & 7);
                const int found_i = (fitm != 0) & (nmy >= 2);                          // new_y < 0: discard (:462-463)
                const bool found = found_i != 0;
                int nul = (isT && dir != 2 && kidx == nk - 1) ? 1 : 0;
                uint32_t nbit = 1u << nmx;
                const uint32_t tv = S.visited[nrot][nmy];
                bool tvis = found & ((tv & nbit) != 0);
                bool push = found && !tvis;
                uint64_t pm = __ballot(push);
                if (pm) {
                    int pos = tail + lane_rank(pm, lane);
                    if (push && pos < QCAP)
                        S.queue[pos] = (uint16_t)(nmx | (nmy << 4) | (nrot << 10) | (1 << 12) | (nul << 13));
                    tail += __popcll(pm);
                    if (tail > QCAP) { *overflow = 1; tail = QCAP; }
                }
                if (isT) {                                                                  // :474-480
                    const uint32_t tb = vrow[nmy + 1 < MAP_H ? nmy + 1 : nmy];
                    const bool tstuck = (nmy + 1 >= MAP_H) | ((tb & nbit) == 0);
                    emit_flagged(S, found && tvis && tstuck, nrot, nmy, nbit, nul, lane, true);
                }
            }
            wave_sync();
            PROF_T(q4); PROF_ADD(4, q3, q4); PROF_ADD(10, 0ull, (unsigned long long)((n_tasks + 63) / 64));
        }
        PROF_T(c0);

        if (stopped) total = -1;
        else {
            total = placements_to_moves(S, type, out, out_cap, lane);
            if (total > out_cap) { *overflow = 1; total = out_cap; }
        }
        PROF_T(c1); PROF_ADD(5, c0, c1);
    }
    wave_sync();
    PROF_T(pz); PROF_ADD(6, p0, pz); PROF_ADD(11, 0ull, 1ull);
#ifdef TRL_PROFILE
    if (lane == 0) for (int i = 0; i < 12; i++) atomicAdd(&g_prof[i], prof_acc[i]);
#endif
    return total;
}

// ---- the same search as a closure on column masks (round 3) ---------------------------------------------------------
// The reference's generator is a FIFO walk, but for every piece other than T its RESULT does not depend on the walk's
// order: the visited set is the closure of the start cell under {left, right, down} inside a rotation state
// (move_generation.py:409-425, 485-488) and under the kick edges of EDGE cells (:427-481) -- a static graph: which kick of a
// list fits a cell first, and where it lands, depends on the validity maps only -- and the placements are its cells that
// cannot move down (:439-440).  For T the order decides one thing: the sub-plane of a stuck cell whose flagged arrivals came
// BOTH through the last kick of a list and through another kick (the last flagged writer wins, :671-677).  A T search with
// no such cell is order-free as well; one with such a cell takes the ordered walk above (`*need_ordered`; 21 % of the T
// searches of the fixture games, 3 % of all searches).  scripts/bfs_closure_proto.py states this in Python and checks it
// against the reference's lists (0 differences on 1 884 + 360 fixture searches outside those T cases).
//
// Layout: one lane per (rotation, map column): rot = lane >> 4, mx = lane & 15 (14 columns; lanes 14, 15 of a row idle);
// a lane holds 64-bit COLUMN masks over the 44 map rows.  "Down" is a carry chain (one add), left / right are DPP row
// shifts by one lane, so one step of the fixed point is ~20 vector instructions for all four rotations at once and touches
// no memory; kicks go through LDS (read the target column's validity, OR the arrivals into the target lane's word).
// The ordered walk needs 2 700 vector instructions and 376 dependent LDS round trips per search; this one ~1 200 and ~40.
struct alignas(8) CsScratch {
    uint16_t present[4][MAP_H];      // the three planes placements_to_moves reads (same names as MgScratch)
    uint16_t flagp[4][MAP_H];
    uint16_t ulp[4][MAP_H];
    uint64_t fcol[20];               // free cells of board column x (bit = map row) at index x + 2, zero elsewhere
    uint64_t vcol[4 * 20];           // valid[rot] column mx at index rot * 20 + mx + 2; the other entries are zero guards
    uint64_t arr[2][4 * 20];         // kick arrivals, same indexing: [1] through the last kick of a list (T, not 180), [0] others
    uint64_t kick[12];               // this piece's kick lists, [rot * 3 + dir - 1] (six registers less across the fixed point)
    uint64_t snap[64];               // T: per lane, the cells reached after generation 0 (rotation 0) / 1 (the others): resolve_mixed_wave
};
static_assert(sizeof(CsScratch) <= sizeof(MgScratch), "the closure search's scratch shares the ordered walk's memory");

__device__ __forceinline__ uint64_t dpp64(uint64_t v, const int ctrl_is_shl) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    if (ctrl_is_shl) {      // row_shl:1 -- lane i takes lane i + 1 of its row of 16 (zero at the end of the row)
        lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x101, 0xf, 0xf, false);
        hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x101, 0xf, 0xf, false);
    } else {                // row_shr:1 -- lane i takes lane i - 1
        lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x111, 0xf, 0xf, false);
        hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x111, 0xf, 0xf, false);
    }
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// column mask -> one bit per set row in a plane of row masks (LDS atomics on the 32-bit words holding two rows)
__device__ __forceinline__ void scatter_column(uint16_t *plane, uint64_t col, int rot, int mx) {
    unsigned int *w = reinterpret_cast<unsigned int *>(plane);
    while (__ballot(col != 0)) {
        if (col) {
            const int my = __ffsll((unsigned long long)col) - 1;
            col &= col - 1;
            const int idx = rot * MAP_H + my;
            atomicOr(w + (idx >> 1), (1u << mx) << ((idx & 1) * 16));
        }
    }
}

// (butterfly through ds_bpermute with the caller's lane index: __shfl_xor derives the lane from mbcnt, a value the compiler
// then keeps for the whole launch)
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v, int lane) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const int src = (lane ^ off) << 2;
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)v);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(v >> 32));
        const uint64_t o = (uint64_t)lo | ((uint64_t)hi << 32);
        v = o > v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v, int lane) {
#pragma unroll
    for (int off = 32; off; off >>= 1) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ off) << 2, (int)v);
        v = o < v ? o : v;
    }
    return v;
}

// Mixed T cells (flagged arrivals through a last kick AND through another kick: the last flagged writer decides the
// sub-plane, move_generation.py:671-677) whose last writer follows from the first two generations of the walk, without
// walking.  The FIFO's order is the order of the pushes; the spawn region (generation 0, rotation 0) is one flood, so the
// generation-1 items are ordered by the (row, column, direction) of their source cell in it, and for the cells of the
// other three rotation states that the FIRST item pushed into their state floods (nearly all of a state's generation-1
// cells; pockets that only later items reach are the exception), the time of a flagged write is a tuple of map coordinates:
//   arrival from a generation-0 source on a cell first visited in generation 1: written when the pushed item is popped
//       (1, key(source), 0)                                               key = (row, column, direction)
//   arrival from a generation-1 source s on a cell already visited by then -- a cell of the spawn region, or of a state
//   whose first item was pushed before s's state's -- written while s's state is flooded
//       (1, seedkey(rot s), 1, key(s))                                    seedkey = key of the first push into the state
//   ... on a cell not visited yet: pushed, written when popped in generation 2:   (2, seedkey(rot s), key(s))
// The latest write wins.  A search with a mixed cell outside these cases (the cell or one of its sources first reached in a
// later generation, or outside its state's first flood) is left to the ordered walk (returns false): 3.8 % of the T
// searches of the fixture games instead of 21 %.  scripts/bfs_closure_proto.py: resolve_mixed, checked against every fixture list.
// C.snap holds the generation snapshots; the three planes (not written yet) lend their memory: the caller has left every
// lane's reached edge cells in the first 64 words (and what it wants to keep in the next 64).  It starts from LDS alone --
// nothing of the closure search stays in registers across it (movegen_closure_finish_mixed): the registers of this rare
// path (a fifth of the T searches) must not weigh on the allocation of the kernels' ply loops.
static __device__ __forceinline__ bool resolve_mixed_wave(CsScratch &C, int lane) {
    const int rot = lane >> 4, mx = lane & 15;
    uint64_t *reL = reinterpret_cast<uint64_t *>(&C.present[0][0]);                     // [64], 512 of the planes' 1056 bytes
    const uint64_t valid = C.vcol[rot * 20 + mx + 2], re = reL[lane];
    uint64_t mixed;
    {
        const uint64_t stuck = valid & ~(valid >> 1);
        mixed = C.arr[1][rot * 20 + mx + 2] & C.arr[0][rot * 20 + mx + 2] & stuck;
    }
    // the first push into each of the states 1..3: the sources are the spawn region's edge cells, in (row, column, dir) order
    uint32_t seed[3];                                                                    // key << 16 | target column << 8 | target row
    {
        const uint64_t src0 = (rot == 0) ? (C.snap[lane] & re) : 0ull;
#pragma unroll
        for (int d = 0; d < 3; d++) {
            uint64_t kq = C.kick[d];
            const int nk = (int)(kq >> 60);
            uint64_t rem = src0;
            uint32_t best = 0xFFFFFFFFu;
#pragma unroll 1
            for (int ki = 0; ki < 6; ki++, kq >>= 6) {
                if (!__ballot(rem != 0 && ki < nk)) break;
                const int fx = (int)(kq & 7), fy = (int)((kq >> 3) & 7);
                const uint64_t vt = C.vcol[(d + 1) * 20 + mx + fx];
                uint64_t fit = (vt << 2) >> (4 - fy);
                if (ki >= nk) fit = 0;
                uint64_t moved = rem & fit;
                rem &= ~fit;
                moved &= ~((1ull << fy) - 1ull);                                         // target row my - ky >= 2, else discarded (:462-463)
                if (moved) {
                    const int my = __ffsll((unsigned long long)moved) - 1;
                    const uint32_t cand = ((uint32_t)((my << 6) | (mx << 2) | d) << 16) | ((uint32_t)(mx + fx - 2) << 8) | (uint32_t)(my - (fy - 2));
                    best = cand < best ? cand : best;
                }
            }
            seed[d] = wave_min_u32(best, lane);
        }
    }
    // the flood of the first item pushed into each of the states 1..3
    {
        uint64_t dm = 0;
        if (rot > 0) {
            const uint32_t sd = rot == 1 ? seed[0] : rot == 2 ? seed[1] : seed[2];
            if (sd != 0xFFFFFFFFu && (int)((sd >> 8) & 0xFF) == mx) dm = 1ull << (sd & 0xFF);
        }
        for (;;) {
            const uint64_t old = dm;
            dm |= ((valid + dm) ^ valid) & valid;
            dm |= (dpp64(dm, 0) | dpp64(dm, 1)) & valid;
            if (!__ballot(dm != old)) break;
        }
        // A state's generation-1 cells may have more pieces than this first flood (pockets that later items reach): the
        // rule needs only the cells it looks at -- the mixed cell and its sources -- to lie in the first flood, so from
        // here on the snapshot of the states 1..3 IS the first flood (a subset of the generation-1 cells)
        if (rot > 0) C.snap[lane] = dm;
        wave_sync();
    }
    // cell by cell: the candidate sources (three other states x up to six kicks) one per lane
    bool ok = true;
    for (;;) {
        const uint64_t have = __ballot(mixed != 0);
        if (!have) break;
        const int L = __ffsll((unsigned long long)have) - 1;
        const int rK = L >> 4, xK = L & 15;
        const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mixed, L), whi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mixed >> 32), L);
        const uint64_t word = (uint64_t)wlo | ((uint64_t)whi << 32);
        const int yK = __ffsll((unsigned long long)word) - 1;
        if (lane == L) mixed &= mixed - 1;
        if (!((C.snap[L] >> yK) & 1)) { ok = false; break; }                              // the cell itself: generation <= 1 (0 in the spawn state)
        uint64_t t = 0;                                                                  // (time << 1) | last kick, 0: no arrival
        bool deep = false;
        if (lane < 18) {
            const int rs = (rK + 1 + lane / 6) & 3, ki = lane % 6;
            const int d = (rK - rs - 1) & 3;
            const uint64_t kq = C.kick[rs * 3 + d];
            const int nk = (int)(kq >> 60);
            if (ki < nk) {
                const int fx = (int)((kq >> (6 * ki)) & 7), fy = (int)((kq >> (6 * ki + 3)) & 7);
                const int sx = xK - (fx - 2), sy = yK + (fy - 2);
                if (sx >= 0 && sx < MAP_W && sy >= 0 && sy < MAP_H && ((reL[rs * 16 + sx] >> sy) & 1)) {
                    bool first = true;                                                   // no earlier kick of the list fits there
                    for (int j = 0; j < ki; j++) {
                        const int fx2 = (int)((kq >> (6 * j)) & 7), fy2 = (int)((kq >> (6 * j + 3)) & 7);
                        const uint64_t vt = C.vcol[rK * 20 + sx + fx2];
                        if ((((vt << 2) >> (4 - fy2)) >> sy) & 1) first = false;
                    }
                    if (first) {
                        const uint64_t skey = (uint64_t)((sy << 6) | (sx << 2) | d);
                        const uint64_t ulk = (d != 1 && ki == nk - 1) ? 1ull : 0ull;
                        if (!((C.snap[rs * 16 + sx] >> sy) & 1)) deep = true;            // a source of a later generation
                        else if (rs == 0) t = (((1ull << 40) | (skey << 21)) << 1) | ulk;
                        else {
                            const uint64_t ks = (rs == 1 ? seed[0] : rs == 2 ? seed[1] : seed[2]) >> 16;
                            const uint64_t kK = rK == 0 ? 0ull : (rK == 1 ? seed[0] : rK == 2 ? seed[1] : seed[2]) >> 16;
                            const bool visited = rK == 0 || kK < ks;
                            t = ((visited ? ((1ull << 40) | (ks << 21) | (1ull << 20) | skey) : ((2ull << 40) | (ks << 21) | skey)) << 1) | ulk;
                        }
                    }
                }
            }
        }
        if (__ballot(deep)) { ok = false; break; }
        const uint64_t win = wave_max_u64(t, lane);
        if (win == 0) { ok = false; break; }                                             // no arrival found: not this rule's case
        if (lane == L) {
            unsigned long long *w = reinterpret_cast<unsigned long long *>(&C.arr[(win & 1) ? 0 : 1][rK * 20 + xK + 2]);
            *w &= ~(1ull << yK);                                                         // the losing class leaves the cell
        }
    }
    wave_sync();
    return ok;
}

// a 64-bit zero the compiler cannot keep: as an ordinary constant it is hoisted out of the kernels' ply loops as a register
// pair, spilled there, and re-loaded from scratch memory for every guard word below
__device__ __forceinline__ uint64_t fresh_zero64() {
    uint32_t lo, hi;
    asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(lo), "=v"(hi));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// returns the number of placements, or -2: a T search with mixed cells (movegen_closure_finish_mixed goes on from here)
static __device__ __forceinline__ int movegen_closure_wave(const uint16_t *rows, int type, bool need_spawn, CsScratch &C,
                                                           uint16_t *out, int out_cap, int lane, int *overflow) {
    const int rot = lane >> 4, mx = lane & 15;
    const bool isT = (type == PT), check_rot = (type != PO);
#ifdef TRL_PROFILE
    unsigned long long cp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define CPROF_FLUSH() do { if (lane == 0) for (int i_ = 0; i_ < 12; i_++) atomicAdd(&g_cprof[i_], cp[i_]); } while (0)
#define CPROF_ADD(i, a, b) do { cp[i] += (b) - (a); } while (0)
#else
#define CPROF_FLUSH()
#define CPROF_ADD(i, a, b)
#endif
    PROF_T(c_0);
    // free cells per board column: ten ballots over the rows (lane = board row), column x kept by lane x + 2
    const uint32_t myrow = lane < TRL_ROWS ? (uint32_t)rows[lane < TRL_ROWS ? lane : 0] : 0x3FFu;
    uint64_t mine = 0;
#pragma unroll
    for (int x = 0; x < TRL_COLS; x++) {
        const uint64_t occ = __ballot((myrow >> x) & 1u);
        const uint64_t fr = (~occ & ((1ull << TRL_ROWS) - 1ull)) << 2;
        if (lane == x + 2) mine = fr;
    }
    if (lane < 20) C.fcol[lane] = mine;
    if (lane < 12) C.kick[lane] = C_KICK[((type == PI) ? 12 : 0) + lane];
    {
        uint32_t *z = reinterpret_cast<uint32_t *>(&C.arr[0][0]);
        for (int i = lane; i < 2 * 4 * 20 * 2; i += 64) z[i] = 0;
        uint32_t *pz = reinterpret_cast<uint32_t *>(&C.present[0][0]);
        for (int i = lane; i < (isT ? 3 : 1) * 4 * MAP_H / 2; i += 64) pz[i] = 0;
        if (isT) C.snap[lane] = fresh_zero64();
    }
    // _get_highest_row / _set_starting_position (move_generation.py:164-180)
    const uint64_t occm = __ballot(lane < TRL_ROWS && myrow != 0);
    const int highest = occm ? (__ffsll((unsigned long long)occm) - 1) : TRL_ROWS;
    int sy = highest - C_BOX[type]; if (sy < SPAWN_Y) sy = SPAWN_Y;
    const int sx = spawn_x(type);
    wave_sync();
    // validity of (rot, mx): all four minos on free cells (move_generation.py:490-528), as a column
    uint64_t valid = (1ull << MAP_H) - 1ull;
    {
        const uint32_t m = C_MINO[type * 4 + rot];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cx = (int)((m >> (4 * i)) & 3), cy = (int)((m >> (4 * i + 2)) & 3);
            valid &= C.fcol[mx + cx] >> cy;
        }
        if (mx >= MAP_W) valid = 0;
    }
    C.vcol[rot * 20 + mx + 2] = valid;                      // (mx 14, 15: zeros at 16, 17)
    if (mx < 2) C.vcol[rot * 20 + mx] = fresh_zero64();                  // guards 0, 1
    if (mx >= MAP_W) C.vcol[rot * 20 + mx + 4] = fresh_zero64();         // guards 18, 19
    wave_sync();
    const uint64_t v0 = C.vcol[sx + 2 + 2];                 // rotation 0, column sx + 2
    bool ok = (v0 >> (sy + 2)) & 1;                                                      // :349-351
    if (need_spawn) ok = ok && ((v0 >> (SPAWN_Y + 2)) & 1);                              // player.py:40
    if (!uni(ok)) return 0;

    // edge cells (:427-431): cannot move down, or a side neighbour is not valid.  (valid >> 1: bit my = the cell below is valid)
    const uint64_t edge = valid & (~(valid >> 1) | ~dpp64(valid, 0) | ~dpp64(valid, 1));
    uint64_t reach = (lane == sx + 2) ? (1ull << (sy + 2)) : 0ull;
    uint64_t done = 0;
    PROF_T(c_1); CPROF_ADD(0, c_0, c_1); CPROF_ADD(8, 0ull, 1ull);
    for (int round = 0;; round++) {
        PROF_T(c_f0);
        // fixed point under down / left / right inside every rotation state
        for (;;) {
            const uint64_t old = reach;
            reach |= ((valid + reach) ^ valid) & valid;                                  // down: carry chain through the valid run
            reach |= (dpp64(reach, 0) | dpp64(reach, 1)) & valid;                        // one column to either side
            CPROF_ADD(9, 0ull, 1ull);
            if (!__ballot(reach != old)) break;
        }
        PROF_T(c_f1); CPROF_ADD(1, c_f0, c_f1);
        if (!check_rot) break;
        // T: the cells of generation 0 (the spawn region, rotation 0) and of generation <= 1 (resolve_mixed_wave)
        if (isT && round < 2 && (round == 0) == (rot == 0)) C.snap[lane] = reach;
        // kicks of the edge cells that have not sent theirs yet (each cell sends once, like the flood that first visits it)
        const uint64_t src = reach & edge & ~done;
        done |= src;
        if (!__ballot(src != 0)) break;
        PROF_T(c_k0); CPROF_ADD(10, 0ull, 1ull);
#ifdef TRL_PROFILE
        {   // how many source cells does this round have (a round costs the same whether 1 or 100)
            const int cells = __builtin_amdgcn_readlane(wave_incl_scan(__popcll(src), lane), 63);
            if (lane == 0) {
                atomicAdd(&g_cprof[12], (unsigned long long)cells);
                if (cells <= 2) atomicAdd(&g_cprof[13], 1ull);
                else if (cells <= 6) atomicAdd(&g_cprof[14], 1ull);
                else if (cells <= 12) atomicAdd(&g_cprof[15], 1ull);
                if (round >= 2) atomicAdd(&g_cprof[5], 1ull);
                if (round >= 2 && cells <= 6) atomicAdd(&g_cprof[6], 1ull);
            }
        }
#endif
#pragma unroll
        for (int d = 0; d < 3; d++) {
            const int nrot = (rot + d + 1) & 3;
            uint64_t kq = C.kick[rot * 3 + d];
            const int nk = (int)(kq >> 60);
            uint64_t rem = src;
#pragma unroll 1
            for (int ki = 0; ki < 6; ki++, kq >>= 6) {
                if (!__ballot(rem != 0 && ki < nk)) break;
                CPROF_ADD(11, 0ull, 1ull);
                const int fx = (int)(kq & 7), fy = (int)((kq >> 3) & 7);                 // kick (fx - 2, fy - 2)
                const int idx = nrot * 20 + mx + fx;                                     // target column mx + kx, + 2
                const uint64_t vt = C.vcol[idx];
                uint64_t fit = (vt << 2) >> (4 - fy);                                    // bit my: the target row my - ky is valid
                if (ki >= nk) fit = 0;
                const uint64_t moved = rem & fit;                                        // first fitting kick wins (:452-470)
                rem &= ~fit;
                const uint64_t dest = ((moved << 2) >> fy) & ~3ull;                      // rows my - ky; new_y < 0: discarded (:462-463)
                if (dest) {
                    const int last = (isT && d != 1 && ki == nk - 1) ? 1 : 0;
                    atomicOr(reinterpret_cast<unsigned long long *>(&C.arr[last][idx]), (unsigned long long)dest);
                }
            }
        }
        wave_sync();
        const uint64_t add = (C.arr[0][rot * 20 + mx + 2] | C.arr[1][rot * 20 + mx + 2]) & ~reach;
        reach |= add;
        PROF_T(c_k1); CPROF_ADD(2, c_k0, c_k1);
        if (!__ballot(add != 0)) break;
    }
    PROF_T(c_2);
    // placements: cells that cannot move down (:439-440); T: flagged arrivals on such cells (:386-392, 474-480)
    const uint64_t stuck = valid & ~(valid >> 1);
    if (isT) {
        const uint64_t f1 = C.arr[1][rot * 20 + mx + 2] & stuck, f0 = C.arr[0][rot * 20 + mx + 2] & stuck;
        if (__ballot((f1 & f0) != 0)) {
            // the walk's order decides these cells: movegen_closure_finish_mixed settles them from the first two generations,
            // or the walk is taken.  The planes' memory (zero so far) holds what that stage starts from
            uint64_t *keep = reinterpret_cast<uint64_t *>(&C.present[0][0]);
            keep[lane] = reach & edge;
            keep[64 + lane] = reach & stuck;
            wave_sync();
            CPROF_FLUSH();
            return -2;
        }
        scatter_column(&C.flagp[0][0], f1 | f0, rot, mx);
        scatter_column(&C.ulp[0][0], f1, rot, mx);
    }
    scatter_column(&C.present[0][0], reach & stuck, rot, mx);
    wave_sync();
    int total = placements_to_moves(C, type, out, out_cap, lane);
    if (total > out_cap) { *overflow = 1; total = out_cap; }
    wave_sync();
    PROF_T(c_3); CPROF_ADD(3, c_2, c_3);
    CPROF_FLUSH();
    return total;
}

// second stage of a T search that stopped at mixed cells (movegen_closure_wave returned -2): settle them, then the planes and
// the move list as in the first stage; -1: the ordered walk is needed
static __device__ __forceinline__ int movegen_closure_finish_mixed(CsScratch &C, uint16_t *out, int out_cap, int lane, int *overflow) {
    const int rot = lane >> 4, mx = lane & 15;
    if (!uni(resolve_mixed_wave(C, lane))) return -1;
    const uint64_t placed = reinterpret_cast<uint64_t *>(&C.present[0][0])[64 + lane];
    const uint64_t valid = C.vcol[rot * 20 + mx + 2], stuck = valid & ~(valid >> 1);
    const uint64_t f1 = C.arr[1][rot * 20 + mx + 2] & stuck, f0 = C.arr[0][rot * 20 + mx + 2] & stuck;
    wave_sync();
    uint32_t *pz = reinterpret_cast<uint32_t *>(&C.present[0][0]);
    for (int i = lane; i < 3 * 4 * MAP_H / 2; i += 64) pz[i] = 0;
    wave_sync();
    scatter_column(&C.flagp[0][0], f1 | f0, rot, mx);
    scatter_column(&C.ulp[0][0], f1, rot, mx);
    scatter_column(&C.present[0][0], placed, rot, mx);
    wave_sync();
    int total = placements_to_moves(C, PT, out, out_cap, lane);
    if (total > out_cap) { *overflow = 1; total = out_cap; }
    wave_sync();
    return total;
}

#ifndef TRL_CLOSURE
#define TRL_CLOSURE 1       // 0: every search takes the ordered walk (A/B builds)
#endif
static __device__ __forceinline__ int movegen_wave(const uint16_t *rows, int type, bool need_spawn, MgScratch &S,
                                                   uint16_t *out, int out_cap, int lane, int *overflow) {
#if TRL_CLOSURE
    int n = movegen_closure_wave(rows, type, need_spawn, reinterpret_cast<CsScratch &>(S), out, out_cap, lane, overflow);
    if (n >= 0) return n;
#ifndef TRL_NO_RESOLVE_MIXED
    if (n == -2) {
        n = movegen_closure_finish_mixed(reinterpret_cast<CsScratch &>(S), out, out_cap, lane, overflow);
        if (n >= 0) return n;
    }
#endif
#endif
    return movegen_ordered_wave(rows, type, need_spawn, S, out, out_cap, lane, overflow);
}

// ---- the reference's default generator, 'brute-force' (move_generation.py:204-243) ------------------------------
// A cell-by-cell FIFO search: pop a pose; already checked -> (if it cannot move down and this exact (pose, flags) is
// not in the placement list yet, append it) and go on; else mark, append if it cannot move down, push right, left,
// down (flags cleared), then the three rotations through Player.try_wallkick (player.py:77-102: first fitting kick;
// only T sets the flags; new_y < 0 discarded).  The result differs from 'convolutional' only in which T-spin
// sub-plane a placement lands in (the order in which flagged arrivals reach a cell).
// Not a hot path (tests, get_move_matrix(player) with the reference's default argument): lane 0 of the wavefront
// walks the queue, the lanes share the validity maps and the conversion to moves.
constexpr int BF_QCAP = 10304;       // pushes happen on first visits only: 1 + 6 * (4 * 39 * 11) poses at most
struct alignas(4) BfScratch {
    uint16_t valid[4][MAP_H];
    uint16_t checked[4][MAP_H];      // checked_list
    uint16_t present[4][MAP_H];      // an unflagged placement is in the list
    uint16_t flagp[4][MAP_H];        // a flagged one is
    uint16_t ulp[4][MAP_H];          // the LAST flagged one appended used the last kick ("last flagged writer wins", :671-677)
    uint16_t seen_fn[4][MAP_H];      // (flagged, no last kick) is in the list  -- `piece_location not in place_location_queue`
    uint16_t seen_fl[4][MAP_H];      // (flagged, last kick) is in the list        compares the flags too (dataclass equality)
    uint16_t queue[BF_QCAP];         // mx:4 | my:6 | rot:2 | flag:1 | last_kick:1
};

static __device__ __forceinline__ int movegen_bruteforce_wave(const uint16_t *rows, int type, bool need_spawn, BfScratch &S,
                                                              uint16_t *out, int out_cap, int lane, int *overflow) {
    for (int i = lane; i < 4 * MAP_H; i += 64) {
        int r = i / MAP_H, ym = i - r * MAP_H;
        S.valid[r][ym] = (uint16_t)valid_row(rows, type, r, ym);
        S.checked[r][ym] = 0; S.present[r][ym] = 0; S.flagp[r][ym] = 0; S.ulp[r][ym] = 0; S.seen_fn[r][ym] = 0; S.seen_fl[r][ym] = 0;
    }
    uint64_t occ = __ballot(lane < TRL_ROWS && rows[lane < TRL_ROWS ? lane : 0] != 0);
    int highest = occ ? (__ffsll((unsigned long long)occ) - 1) : TRL_ROWS;
    int sy = highest - C_BOX[type]; if (sy < SPAWN_Y) sy = SPAWN_Y;
    const int sx = spawn_x(type);
    wave_sync();
    bool ok = (S.valid[0][sy + 2] >> (sx + 2)) & 1;
    if (need_spawn) ok = ok && ((S.valid[0][SPAWN_Y + 2] >> (sx + 2)) & 1);
    ok = uni(ok);
    int total = 0;
    if (ok) {
        if (lane == 0) {
            auto fits = [&](int r, int mx, int my) -> bool {
                return (unsigned)my < (unsigned)MAP_H && (unsigned)mx < 14u && ((S.valid[r][my] >> mx) & 1);
            };
            auto append = [&](int r, int mx, int my, int fl, int ul) {
                const uint16_t bit = (uint16_t)(1u << mx);
                if (!fl) S.present[r][my] |= bit;
                else {
                    S.flagp[r][my] |= bit;
                    if (ul) { S.seen_fl[r][my] |= bit; S.ulp[r][my] |= bit; }
                    else { S.seen_fn[r][my] |= bit; S.ulp[r][my] &= (uint16_t)~bit; }
                }
            };
            const bool check_rot = (type != PO), isT = (type == PT);
            const int ktab = (type == PI) ? 12 : 0;
            int head = 0, tail = 1;
            S.queue[0] = (uint16_t)((sx + 2) | ((sy + 2) << 4));
            while (head < tail) {
                const uint32_t it = S.queue[head++];
                const int mx = it & 15, my = (it >> 4) & 63, rot = (it >> 10) & 3, fl = (it >> 12) & 1, ul = (it >> 13) & 1;
                const uint16_t bit = (uint16_t)(1u << mx);
                const bool down_ok = fits(rot, mx, my + 1);
                if (S.checked[rot][my] & bit) {                                         // :212-217
                    if (!down_ok) {
                        const uint16_t have = !fl ? S.present[rot][my] : (ul ? S.seen_fl[rot][my] : S.seen_fn[rot][my]);
                        if (!(have & bit)) append(rot, mx, my, fl, ul);
                    }
                    continue;
                }
                S.checked[rot][my] |= bit;
                if (!down_ok) append(rot, mx, my, fl, ul);                              // :221-222
                auto push = [&](int nmx, int nmy, int nrot, int nfl, int nul) {
                    if (tail < BF_QCAP) S.queue[tail++] = (uint16_t)(nmx | (nmy << 4) | (nrot << 10) | (nfl << 12) | (nul << 13));
                    else *overflow = 1;
                };
                if (fits(rot, mx + 1, my)) push(mx + 1, my, rot, 0, 0);                 // :225-232, in this order
                if (fits(rot, mx - 1, my)) push(mx - 1, my, rot, 0, 0);
                if (down_ok) push(mx, my + 1, rot, 0, 0);
                if (check_rot)
                    for (int dir = 1; dir <= 3; dir++) {                                // :235-243 + player.py:77-102
                        const int nrot = (rot + dir) & 3;
                        const uint64_t kt = C_KICK[ktab + rot * 3 + dir - 1];
                        const int nk = (int)(kt >> 60);
                        for (int ki = 0; ki < nk; ki++) {
                            const int tx = mx - 2 + (int)((kt >> (6 * ki)) & 7), ty = my + 2 - (int)((kt >> (6 * ki + 3)) & 7);
                            if (!fits(nrot, tx, ty)) continue;
                            if (ty >= 2)                                                // new_location.y >= 0
                                push(tx, ty, nrot, isT ? 1 : fl, isT ? ((ki == nk - 1 && dir != 2) ? 1 : 0) : ul);
                            break;
                        }
                    }
            }
        }
        wave_sync();
        total = placements_to_moves(S, type, out, out_cap, lane);
        if (total > out_cap) { *overflow = 1; total = out_cap; }
    }
    wave_sync();
    return total;
}

// piece types the generator looks at (move_generation.py:91-105 + the hold simulation of
// :112-121): slot 0 = the active piece, slot 1 = what a hold would bring (held, else next)
__device__ __forceinline__ void movegen_types(const trl_player_state &p, int *t0, int *t1) {
    int p1 = p.piece;
    int p2 = (p.held != P_NONE) ? p.held : (p.queue_len > 0 ? (int)(p.queue & 15) : P_NONE);
    *t0 = p1;
    *t1 = (p2 != p1) ? p2 : P_NONE;
}

// ---- stats.py:30-157 in integers --------------------------------------------------------------
__device__ __forceinline__ void update_b2b_level(trl_player_state &p) {      // stats.py:30-43
    int b = p.b2b, lvl = p.b2b_level, t;
    t = (b >= 1370) ? 8 : (b >= 504) ? 7 : (b >= 185) ? 6 : (b >= 67) ? 5 : (b >= 24) ? 4 : (b >= 8) ? 3 : (b >= 3) ? 2 : (b >= 1) ? 1 : 0;
    if (b >= -1 && lvl < t) lvl = t;
    p.b2b_level = (uint8_t)lvl;
}
// floor(q(1 + c/4)) style terms are evaluated as integer products over 4 or 8
static __device__ __forceinline__ int get_attack(trl_player_state &p, int ruleset, int rows, bool tspin, bool mini, bool all_clear) {
    int attack = 0;
    if (rows == 0) {
        p.combo = 0;
    } else {
        int c4;      // 4 + combo
        if (ruleset == TRL_RULESET_S1) {                                      // stats.py:49-86
            bool is_b2b = (rows == 4) || tspin;
            p.b2b = is_b2b ? (int16_t)(p.b2b + 1) : (int16_t)-1;
            update_b2b_level(p);
            int lvl = p.b2b_level; c4 = 4 + p.combo;
            if (rows == 1) {
                if (!tspin) { p.b2b = -1; attack += (p.combo + 2) >> 2; }
                else if (mini) {
                    if (p.b2b <= 0 && lvl <= 0) attack += (p.combo + 2) >> 2;
                    else attack += (lvl * c4) >> 2;
                } else attack += ((2 + lvl) * c4) >> 2;
            } else {
                int a2 = tspin ? (mini ? rows : 4 * rows) : 0;               // 2 * (2 rows tspin (1 - 3/4 mini))
                int p2 = 2 * ((tspin ? 0 : (1 << (rows - 2))) + ((tspin || rows == 4) ? lvl : 0));
                attack += (c4 * (a2 + p2)) >> 3;
            }
            p.combo += 1;
            if (all_clear) attack += 10;
        } else {                                                               // 's2', stats.py:88-133
            bool is_b2b = tspin || mini || rows == 4;
            if (is_b2b) p.b2b += 1;
            else if (all_clear) { attack += 5; p.b2b += 1; }
            else { if (p.b2b >= 4) attack += p.b2b; p.b2b = -1; }
            update_b2b_level(p);
            int lvl = p.b2b_level > 0 ? 1 : 0; c4 = 4 + p.combo;
            if (rows == 1) {
                if (!tspin) attack += (p.combo + 2) >> 2;
                else if (mini) {
                    if (p.b2b <= 0 && p.b2b_level <= 0) attack += (p.combo + 2) >> 2;
                    else attack += (lvl * c4) >> 2;
                } else attack += 2 + ((lvl * c4) >> 2);
            } else {
                int a2 = tspin ? (mini ? rows : 4 * rows) : 0;
                int p2 = 2 * ((tspin ? 0 : (1 << (rows - 2))) + (is_b2b ? lvl : 0));
                attack += (c4 * (a2 + p2)) >> 3;
            }
            p.combo += 1;
        }
    }
    p.lines_sent += attack;                                                    // stats.py:135-136
    p.lines_cleared += rows;
    return attack;
}

// ---- player.py / game.py, one lane --------------------------------------------------------------
__device__ __forceinline__ void create_piece(trl_player_state &p, int type) {          // player.py:37-44
    if (pose_fits(p.rows, type, spawn_x(type), SPAWN_Y, 0)) p.piece = (uint8_t)type;
    else p.game_over = 1;
}
__device__ __forceinline__ void create_next_piece(trl_player_state &p) {                // player.py:29-35
    if (p.queue_len > 0) {
        int next = (int)(p.queue & 15);
        p.queue >>= 4; p.queue_len--;
        create_piece(p, next);
    } else if (p.held != P_NONE) {
        int next = p.held; p.held = P_NONE;
        create_piece(p, next);
    }
}
__device__ __forceinline__ void hold_piece(trl_player_state &p) {                       // player.py:190-201
    if (p.held == P_NONE) { p.held = p.piece; p.piece = P_NONE; create_next_piece(p); }
    else { int temp = p.held; p.held = p.piece; create_piece(p, temp); }
}
__device__ __forceinline__ void garbage_pop(trl_player_state &p, int n) {               // n x pop(0)
    for (int k = 0; k < n; k++) {
        p.garbage[0] = (p.garbage[0] >> 4) | (p.garbage[1] << 60);
        p.garbage[1] = (p.garbage[1] >> 4) | (p.garbage[2] << 60);
        p.garbage[2] = (p.garbage[2] >> 4) | (p.garbage[3] << 60);
        p.garbage[3] >>= 4;
    }
    p.garbage_len = (uint8_t)(p.garbage_len - n);
}
__device__ __forceinline__ bool garbage_push(trl_player_state &p, int col, int n) {     // extend([col] * n)
    bool ovf = false;
    for (int k = 0; k < n; k++) {
        int i = p.garbage_len;
        if (i >= TRL_MAX_GARBAGE) { ovf = true; break; }
        p.garbage[i >> 4] |= (uint64_t)col << (4 * (i & 15));
        p.garbage_len = (uint8_t)(i + 1);
    }
    return ovf;
}

// Queue.generate_bag + Game.add_bag_to_all (piece_queue.py:17-21, game.py:34-38)
static __device__ __forceinline__ void add_bag_to_all(trl_game_state &g, uint64_t seed) {
    for (int i = 0; i < 2; i++) {
        uint64_t k = g.bag_ctr++;
        uint32_t bag = 0x6543210u;                       // nibble j = j
        for (int a = 6; a >= 1; a--) {
            int b = (int)rng_below(seed, g.stream, TRL_RNG_BAG, 8 * k + (uint64_t)a, (uint32_t)(a + 1));
            uint32_t va = (bag >> (4 * a)) & 15, vb = (bag >> (4 * b)) & 15;
            bag = (bag & ~(15u << (4 * a)) & ~(15u << (4 * b))) | (va << (4 * b)) | (vb << (4 * a));
        }
        trl_player_state &p = g.p[i];
        if (p.queue_len + 7 > TRL_MAX_QUEUE) { g.flags |= TRL_FLAG_OVERFLOW; continue; }
        p.queue |= (uint64_t)bag << (4 * p.queue_len);
        p.queue_len = (uint8_t)(p.queue_len + 7);
    }
}

// Game.setup (game.py:28-32) on a zeroed slot
static __device__ __forceinline__ void game_setup(trl_game_state &g, int ruleset, uint64_t seed, uint32_t stream) {
    uint64_t *w = reinterpret_cast<uint64_t *>(&g);
    for (int i = 0; i < (int)(sizeof(trl_game_state) / 8); i++) w[i] = 0;
    g.stream = stream; g.ruleset = (uint8_t)ruleset;
    for (int i = 0; i < 2; i++) { g.p[i].piece = P_NONE; g.p[i].held = P_NONE; g.p[i].b2b = -1; }
    add_bag_to_all(g, seed);
    for (int i = 0; i < 2; i++) create_next_piece(g.p[i]);
}

// Game.make_move(move, add_bag, add_history=False)  (game.py:40-98, player.py:109-188), one wavefront.
// All 64 lanes call it with the same arguments; `g` lives in LDS.  Rows are spread over the lanes: the
// hard-drop row is a ballot of "piece fits at origin row = lane" (no drop loop), cleared rows are found and
// compacted with one ballot + popcount, pending garbage is inserted by a parallel row shift; lane 0 keeps the
// scalar bookkeeping (hold, attack, queues, spawn).
// `gctr`: running garbage-draw counter to use instead of the state's own (search-time copies
// draw from the real game's stream, as the reference draws from one global generator)
static __device__ __forceinline__ void make_move(trl_game_state &g, uint64_t seed, int move, bool add_bag, int lane,
                                                 uint32_t *gctr = nullptr) {
    trl_player_state &p = g.p[g.turn];
    trl_player_state &o = g.p[1 - g.turn];
    // Game.move_piece (game.py:40-64)
    const int pc = C_PLANE[TRL_MOVE_PLANE(move)];
    int type = pc & 7;
    const int rot = (pc >> 3) & 3, ts = pc >> 5;
    const int x = TRL_MOVE_COL(move) - 2, y = TRL_MOVE_ROW(move);
    if (lane == 0 && (p.piece == P_NONE || type != p.piece)) hold_piece(p);
    wave_sync();
    if (p.game_over) return;                                                    // game.py:70
    type = p.piece;
    bool rot_flag = ts >= 1, last_kick = ts == 2;
    // Player.place_piece (player.py:109-188).  ghost_y (player.py:46-59): rows the piece fits at, one per lane
    const uint64_t vm = __ballot(lane < TRL_ROWS && pose_fits(p.rows, type, x, lane, rot));
    const int place_y = y + (__ffsll((unsigned long long)(~vm >> (y + 1))) - 1);
    if (!((vm >> place_y) & 1)) { if (lane == 0) g.flags |= TRL_FLAG_BADMOVE; wave_sync(); return; }   // not a legal pose
    if (y != place_y) { rot_flag = false; last_kick = false; }                  // :115-118
    bool tspin = false, mini = false, all_clear = false;
    if (type == PT && rot_flag) {                                               // :127-140
        bool f0 = !cell_free(p.rows, x, place_y), f1 = !cell_free(p.rows, x + 2, place_y);
        bool f2 = !cell_free(p.rows, x + 2, place_y + 2), f3 = !cell_free(p.rows, x, place_y + 2);
        tspin = ((int)f0 + (int)f1 + (int)f2 + (int)f3) >= 3;
        bool front = (rot == 0 ? (f0 && f1) : rot == 1 ? (f1 && f2) : rot == 2 ? (f2 && f3) : (f3 && f0));
        if (!front && !last_kick) mini = true;
    }
    if (g.ruleset == TRL_RULESET_S2 && !tspin) {                                // :143-151 (pose at the move's y)
        bool up = y >= 1 && ((vm >> (y - 1)) & 1), down = (vm >> (y + 1)) & 1;
        bool mobile = up || down || pose_fits(p.rows, type, x - 1, y, rot) || pose_fits(p.rows, type, x + 1, y, rot);
        if (!mobile) mini = true;
    }
    if (lane == 0) {                                                            // :153-157 lock
        uint32_t mm = C_MINO[type * 4 + rot];
        for (int i = 0; i < 4; i++) {
            int col = x + (int)((mm >> (4 * i)) & 3), row = place_y + (int)((mm >> (4 * i + 2)) & 3);
            p.rows[row] |= (uint16_t)(1u << col);
        }
        p.piece = P_NONE;
    }
    wave_sync();
    // :161-178 full rows -> delete + zero rows on top; a surviving row r moves down by the full rows below it
    const uint32_t v = lane < TRL_ROWS ? (uint32_t)p.rows[lane < TRL_ROWS ? lane : 0] : 0u;
    const bool full = v == 0x3FFu;
    const uint64_t fm = __ballot(full);
    const int cleared = __popcll(fm);
    if (cleared) {
        all_clear = __ballot(!full && v != 0) == 0;
        wave_sync();
        if (lane < TRL_ROWS && !full) p.rows[lane + __popcll(fm >> (lane + 1))] = (uint16_t)v;
        if (lane < cleared) p.rows[lane] = 0;
        wave_sync();
    }
    if (lane == 0) {
        uint32_t &garb_ctr = gctr ? *gctr : g.garb_ctr;
        int attack = get_attack(p, g.ruleset, cleared, tspin, mini, all_clear);     // :180-186
        p.pieces += 1;
        // garbage_to_send = [column] * attack, then Game.check_garbage (game.py:100-117)
        int send = attack, column = 0;
        if (attack > 0) column = (int)rng_below(seed, g.stream, TRL_RNG_GARBAGE, garb_ctr++, 10);
        int cancel = send < (int)p.garbage_len ? send : (int)p.garbage_len;
        if (cancel > 0) { garbage_pop(p, cancel); send -= cancel; }
        if (send > 0 && garbage_push(o, column, send)) g.flags |= TRL_FLAG_OVERFLOW;
    }
    wave_sync();
    int n = p.garbage_len;
    if (n > 0 && cleared == 0) {                                                // spawn_garbage, board.py:12-21
        if (n > TRL_ROWS) { if (lane == 0) g.flags |= TRL_FLAG_OVERFLOW; n = TRL_ROWS; }
        const uint32_t w = lane < TRL_ROWS ? (uint32_t)p.rows[lane < TRL_ROWS ? lane : 0] : 0u;
        const int c = (int)((p.garbage[(lane & 63) >> 4] >> (4 * (lane & 15))) & 15);
        wave_sync();
        if (lane >= n && lane < TRL_ROWS) p.rows[lane - n] = (uint16_t)w;
        if (lane < n) p.rows[TRL_ROWS - n + lane] = (uint16_t)(0x3FFu & ~(1u << c));
        wave_sync();
        if (lane == 0) { p.garbage[0] = p.garbage[1] = p.garbage[2] = p.garbage[3] = 0; p.garbage_len = 0; }
    }
    wave_sync();
    if (lane == 0) {
        create_next_piece(p);                                                   // game.py:80
        if (add_bag && p.queue_len < 5) add_bag_to_all(g, seed);                // game.py:82-83
        g.turn ^= 1;
        g.ply++;
    }
    wave_sync();
}

__device__ __forceinline__ bool is_terminal(const trl_game_state &g) { return g.p[0].game_over || g.p[1].game_over; }
__device__ __forceinline__ bool no_move(const trl_game_state &g) {
    const trl_player_state &p = g.p[g.turn];
    return p.piece == P_NONE && p.held == P_NONE;
}

}  // namespace trl
