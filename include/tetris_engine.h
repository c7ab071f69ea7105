/*
 * tetris_engine.h -- C ABI of the MI355X self-play rollout engine.
 *
 * The reference (mat-lee/tetris-reinforcement-learning) is pure Python and has no FFI for
 * this path; the seams a replacement has to honour are plain Python callables
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it stands in
 * for (file:line into the reference tree).  The Python mirror of those callables
 * (the tetris-reinforcement-learning_amd package) is implemented only on top of this header.
 *
 * Conventions
 *   - every function returns 0 on success and a negative TRL_E_* code otherwise; no C++
 *     exception crosses the boundary; trl_last_error() gives a text for the calling thread;
 *   - the caller owns every buffer it passes; "host" pointers are ordinary memory,
 *     "dev" pointers are HIP device pointers (e.g. torch tensor .data_ptr());
 *   - an engine handle is thread-compatible, not thread-safe;
 *   - all work is issued on the engine's HIP stream (trl_engine_create / trl_set_stream);
 *     functions with host outputs synchronise that stream before returning, functions
 *     with only device outputs do not.
 *
 * Coordinates follow the reference: (col,row), row 0 = top, row 39 = floor
 * (board.py:23-30); a move is (plane, col, row) with plane from const.py:82-118.
 */
#ifndef TETRIS_ENGINE_H
#define TETRIS_ENGINE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- sizes (const.py:6-12,38,122) ------------------------------------------------------ */
#define TRL_ROWS          40
#define TRL_COLS          10
#define TRL_SPAWN_ROW     23      /* spawn y = ROWS - SPAWN_ROW = 17                       */
#define TRL_PREVIEWS      5
#define TRL_PLANES        27      /* POLICY_SHAPE[0]                                       */
#define TRL_POLICY_ROWS   39      /* POLICY_SHAPE[1] = ROWS - 1                            */
#define TRL_POLICY_COLS   11      /* POLICY_SHAPE[2] = COLS + 2 - 1 (left buffer of 2)     */
#define TRL_POLICY_SIZE   (TRL_PLANES * TRL_POLICY_ROWS * TRL_POLICY_COLS) /* 11583       */
#define TRL_MAX_MOVES     1000    /* const.py:12 MAX_MOVES (full turns)                    */

#define TRL_MAX_LEGAL     512     /* cap on legal placements per state (reference: <=132 seen) */
#define TRL_MAX_QUEUE     16      /* cap on preview queue length (reference: <=12)          */
#define TRL_MAX_GARBAGE   64      /* cap on pending garbage lines per player                */
#define TRL_NO_PIECE      7       /* piece codes 0..6 index "ZLOSIJT" (const.py:68)         */

#define TRL_RULESET_S1    1
#define TRL_RULESET_S2    2

/* ---- error codes ----------------------------------------------------------------------- */
#define TRL_OK             0
#define TRL_E_INVALID     -1      /* bad argument                                          */
#define TRL_E_HIP         -2      /* a HIP runtime call failed (text in trl_last_error)    */
#define TRL_E_NO_DEVICE   -3      /* no gfx950 device visible: the engine has no CPU path  */
#define TRL_E_OVERFLOW    -4      /* a fixed cap (legal moves, queue, garbage) was hit     */
#define TRL_E_STATE       -5      /* call order violated (e.g. step before reset)          */

/* ---- move encoding --------------------------------------------------------------------- */
/* One legal placement = uint16: plane<<10 | row<<4 | col, col is the POLICY column
 * (board col + 2).  Ascending numeric order == numpy.argwhere order of the reference mask
 * (ai.py:1016-1024 get_move_list), which fixes child order and tie-breaks in the search. */
#define TRL_MOVE(plane, row, col)  ((uint16_t)(((plane) << 10) | ((row) << 4) | (col)))
#define TRL_MOVE_PLANE(m)  (((m) >> 10) & 31)
#define TRL_MOVE_ROW(m)    (((m) >> 4) & 63)
#define TRL_MOVE_COL(m)    ((m) & 15)
#define TRL_MOVE_NONE      0xFFFFu

/* ---- canonical game state (also the layout kept in HBM, 320 B per game) ----------------- */
/* One player (player.py:12-27 + stats.py:5-15), bit-packed:                                 */
typedef struct trl_player_state {
    uint16_t rows[TRL_ROWS];  /* rows[r] bit c = cell (r,c) != 0  (board.py:7; the rules and */
                              /* the net only test != 0, ai.py:1364-1366)                     */
    uint64_t garbage[4];      /* garbage_to_receive: nibble i = hole column of line i         */
                              /* (index 0 spawns first, player.py:19)                         */
    uint64_t queue;           /* queue.pieces: nibble i = i-th piece code                     */
    int32_t  pieces;          /* stats.pieces                                                 */
    int32_t  lines_sent;      /* stats.lines_sent                                             */
    int32_t  lines_cleared;   /* stats.lines_cleared                                          */
    int16_t  b2b;             /* stats.b2b (starts at -1)                                     */
    int16_t  combo;           /* stats.combo                                                  */
    uint8_t  b2b_level;       /* stats.b2b_level (monotone, stats.py:30-43)                   */
    uint8_t  piece;           /* active piece code or TRL_NO_PIECE (always at spawn pose)     */
    uint8_t  held;            /* held_piece code or TRL_NO_PIECE                              */
    uint8_t  queue_len;
    uint8_t  garbage_len;
    uint8_t  game_over;
    uint8_t  pad_[2];
} trl_player_state;           /* 144 bytes */

#define TRL_FLAG_OVERFLOW   1u   /* a cap was exceeded in this game; its state is not trustworthy */
#define TRL_FLAG_BADMOVE    2u   /* env_step was given a move that is not legal in this state      */
#define TRL_FLAG_STREAM_WRAP 4u  /* rollout mode: this slot has been set up again 256 times -- episode e runs on RNG    */
                                 /* stream + e * 2^24 (uint32), so from here on it replays the games of its first episodes */

typedef struct trl_game_state {
    trl_player_state p[2];    /* p[0] = Human/colour 0 moves first, p[1] = AI/colour 1 (player.py:253-261) */
    uint32_t stream;          /* RNG stream id of this game (see rng below)                   */
    uint32_t bag_ctr;         /* bags drawn so far (Queue.generate_bag calls, piece_queue.py:17) */
    uint32_t garb_ctr;        /* garbage columns drawn so far (player.py:184-186)             */
    uint32_t ply;             /* placements made so far in this game                          */
    uint32_t pol_ctr;         /* draws of the built-in random policy                          */
    uint32_t episode;         /* auto-resets of this slot (rollout mode)                      */
    uint8_t  turn;            /* game.turn                                                    */
    uint8_t  flags;           /* TRL_FLAG_*                                                   */
    uint8_t  ruleset;         /* TRL_RULESET_*                                                */
    uint8_t  pad_[5];
} trl_game_state;             /* 320 bytes */

/* ---- random numbers ----------------------------------------------------------------------
 * The reference draws from CPython's global Mersenne Twister (piece_queue.py:17-21,
 * player.py:184-186); that stream cannot be split over thousands of concurrent games, so
 * the engine defines its own counter-based generator and the parity harness injects the
 * same generator into the reference (oracle/ref_harness.py).  For 64-bit seed S, stream s,
 * purpose p, counter c:
 *      mix(z) = splitmix64 finaliser of (z + 0x9E3779B97F4A7C15)
 *      u32(S,s,p,c) = mix( mix(S ^ s * 0xD1342543DE82EF95) ^ (p << 56) ^ c ) >> 32
 *      below(n)     = (u32 * n) >> 32
 *   p = 0  bag k  : Fisher-Yates of "ZLOSIJT", for i = 6..1: j = below(i+1) at c = 8k+i, swap(i,j)
 *   p = 1  garbage: column = below(10) at c = garb_ctr
 *   p = 2  built-in uniform policy: index = below(n_legal) at c = pol_ctr
 */
#define TRL_RNG_BAG      0
#define TRL_RNG_GARBAGE  1
#define TRL_RNG_POLICY   2

typedef struct trl_engine trl_engine;

typedef struct trl_config {
    int32_t  n_games;         /* concurrent games held by the engine                          */
    int32_t  ruleset;         /* TRL_RULESET_S1 / TRL_RULESET_S2  (Config.ruleset, ai.py:71)  */
    int32_t  device;          /* HIP device ordinal                                           */
    int32_t  reserved;
    uint64_t seed;            /* S above                                                      */
    void    *stream;          /* hipStream_t to run on, or NULL for the engine's own stream   */
} trl_config;

/* lifetime */
int  trl_engine_create(const trl_config *cfg, trl_engine **out);
void trl_engine_destroy(trl_engine *h);
int  trl_set_stream(trl_engine *h, void *hip_stream);
const char *trl_last_error(void);
const char *trl_version(void);

/* Game(ruleset).setup() for every slot (game.py:8-32): slot g gets stream id
 * stream_ids[g] (host array of n_games), or g when stream_ids == NULL.                      */
int  trl_env_reset(trl_engine *h, const uint32_t *stream_ids);

/* Raw state access: the parity tests compare these bytes with the oracle's; also the
 * way to load arbitrary positions (the reference mutates Game/Player attributes, S8).
 * host buffers of n_games * sizeof(trl_game_state).                                        */
int  trl_env_get_state(trl_engine *h, trl_game_state *host_out);
int  trl_env_set_state(trl_engine *h, const trl_game_state *host_in);
/* device pointer to the n_games states (resident in HBM)                                    */
int  trl_env_state_dev(trl_engine *h, void **dev_out);

/* get_move_matrix(game.players[game.turn], 'convolutional') for every game
 * (move_generation.py:752-790) followed by get_move_list's ordering (ai.py:1016-1024):
 * counts[g] placements, moves[g*TRL_MAX_LEGAL + i] ascending.  Terminal / no-move games
 * give 0.  Results stay in HBM (trl_env_legal_dev); host pointers may be NULL.               */
int  trl_env_legal_moves(trl_engine *h, uint16_t *host_counts, uint16_t *host_moves);
/* get_move_matrix(player, algo) with the reference's algorithm argument (move_generation.py:752-790, 145-151):
 * TRL_ALGO_BRUTE_FORCE is the reference's DEFAULT argument ('brute-force', move_generation.py:204-243, what its own
 * tests call, tests.py:13-58), TRL_ALGO_CONVOLUTIONAL the one Config.move_algorithm defaults to and the search uses
 * (ai.py:84,414).  The two differ only in the T-spin sub-plane of a few placements.  Any other value: TRL_E_INVALID
 * (the reference raises ValueError for an unknown name; its lossy 'faster-but-loss' / 'harddrop' are not built).    */
#define TRL_ALGO_CONVOLUTIONAL 0
#define TRL_ALGO_BRUTE_FORCE   1
int  trl_env_legal_moves_algo(trl_engine *h, int algo, uint16_t *host_counts, uint16_t *host_moves);
int  trl_env_legal_dev(trl_engine *h, void **dev_counts, void **dev_moves);

/* Game.make_move(move, add_bag, add_history=False) for every game (game.py:40-98).
 * moves: host array of n_games encoded moves; TRL_MOVE_NONE leaves a game untouched.
 * A move that is not in the game's legal list sets TRL_FLAG_BADMOVE and is not applied
 * when check_legal != 0 (the reference would corrupt its state or raise).                    */
int  trl_env_step(trl_engine *h, const uint16_t *host_moves, int add_bag, int check_legal);
int  trl_env_step_dev(trl_engine *h, const void *dev_moves, int add_bag, int check_legal);

/* The C2 workload: n_plies times { legal moves -> uniform pick (TRL_RNG_POLICY) ->
 * make_move(add_bag=True) -> Game.setup() again for finished games (is_terminal,
 * game.py:204-209, or ply == max_plies) } fused on the device, no host round trip.
 * steps_out (host, may be NULL) receives the number of make_move calls executed.             */
int  trl_env_rollout_random(trl_engine *h, int n_plies, int max_plies, uint64_t *steps_out);

/* elapsed device time of the last rollout / legal_moves / step kernel, HIP events on the
 * engine's stream (what bench.py's roofline block is computed from)                          */
int  trl_last_kernel_ms(trl_engine *h, float *ms_out);


/* ==== search: MCTS(config, game, net) for every game slot (ai.py:299-659) ========================
 *
 * Every slot runs the reference's sequential PUCT search on its own GPU-resident tree; slots
 * advance in lockstep "ticks", one simulation per slot per tick, so that the leaf evaluations
 * of all slots form one batch for the policy/value net -- the cross-game coalescing of the
 * reference's BatchedEvaluator (ai.py:662-742) without the host in the loop:
 *
 *     trl_mcts_select(h, grids, vec)      selection + Game.copy/make_move + legal moves + game_to_X
 *     <net forward on grids/vec>          -> values[n], policy[n][11583] (softmaxed, ai.py:1318)
 *     trl_mcts_expand(h, values, policy)  expansion, root noise, backup, FPU refresh; on the last
 *                                         simulation of a search: move choice, record, make_move,
 *                                         game end / re-setup (play_game, ai.py:1610-1695)
 *
 * Searches of different slots finish at different ticks (playout-cap randomisation gives 400
 * or 80 simulations at MAX_ITER=160); a slot that finishes starts its next search on the next
 * tick.  Arithmetic follows the reference under numpy >= 2 (float32 PUCT terms below the root,
 * fp64 at the root and for value sums); see DESIGN.md.
 *
 * Extra RNG purposes (counter = ply of the root position unless noted):
 *   p = 3 playout-cap coin, p = 4 temperature pick: uniform double ((u64 >> 11) + 0.5) * 2^-53;
 *   p = 5 root noise, gamma(alpha) by Marsaglia-Tsang on alpha + 1 with the u^(1/alpha) boost,
 *         counter ((ply * 512 + child) * 64 + attempt) * 4 + k.
 *   p = 6  number of random starting moves of a game (np.random.exponential(0.04 * DIRICHLET_S), ai.py:1592-1593):
 *         -scale * log(uniform at c = 0); one more move is played at random while the remainder is > 0
 *   random starting move (random_start, ai.py:1588-1608): a search of ONE simulation without playout cap, noise or
 *         forced playouts, then pick_random_move_by_policy (ai.py:999-1014) = random.choices(moves, priors of the
 *         root children): itertools.accumulate, x = uniform(p = 4, c = ply) * total, bisect over [0, n - 1).
 */
typedef struct trl_mcts_config {      /* Config fields the search reads (ai.py:63-138) */
    int32_t max_iter;                 /* MAX_ITER */
    int32_t training;                 /* Config.training */
    int32_t fpu_reduction;            /* FpuStrategy == 'reduction' (else 'absolute') */
    int32_t use_root_softmax;
    int32_t use_playout_cap;          /* use_playout_cap_randomization */
    int32_t use_dirichlet;            /* use_dirichlet_noise */
    int32_t use_dir_s;                /* use_dirichlet_s */
    int32_t use_forced;               /* use_forced_playouts_and_policy_target_pruning (selection part) */
    int32_t save_all;                 /* Config.save_all */
    int32_t max_nodes;                /* tree capacity per slot (0 = default 65536) */
    int32_t record_capacity;          /* records held on the device between drains (0 = default) */
    int32_t auto_reset;               /* 1: a finished game is set up again on a fresh stream */
    int32_t use_tanh;                 /* Config.use_tanh: value range -1..1 (value_min / mid / max, negate_value, ai.py:198-216) */
    int32_t random_start;             /* Config.use_random_starting_moves: see below */
    int32_t episodes_base;            /* with auto_reset: slot s plays episodes_base + (s < episodes_extra) games, then idles */
    int32_t episodes_extra;           /* (0, 0 = no limit).  Episode e of a slot runs on RNG stream + e * 2^24: at most 256 */
    int32_t move_algorithm;           /* Config.move_algorithm (ai.py:84, 414): TRL_ALGO_CONVOLUTIONAL (default) / TRL_ALGO_BRUTE_FORCE */
    int32_t reserved0;
    double  cpuct, dpuct;             /* CPUCT, DPUCT */
    double  fpu_value;                /* FpuValue */
    double  root_softmax_temp;        /* RootSoftmaxTemp */
    double  temperature;
    double  playout_cap_chance, playout_cap_mult;
    double  dir_alpha, dir_s, dir_expl;   /* DIRICHLET_ALPHA, DIRICHLET_S, DIRICHLET_EXPLORATION */
    double  c_forced;                 /* CForcedPlayout */
} trl_mcts_config;

#define TRL_GRID_FLOATS   (TRL_ROWS * TRL_COLS)        /* 400 */
#define TRL_VEC_FLOATS    105   /* a_pieces 49, a_b2b, a_combo, a_garbage, o_pieces 49, o_b2b, o_combo, o_garbage, color */

/* one saved move or one game end, as handed to the host (trl_mcts_drain_records) */
#define TRL_REC_MOVE      0
#define TRL_REC_GAME_END  1
typedef struct trl_record {
    uint32_t slot;                    /* game slot */
    uint32_t episode;                 /* games finished in this slot before this one */
    uint32_t ply;                     /* ply of the root position (TRL_REC_MOVE) / plies played (GAME_END) */
    uint16_t n_children;
    uint8_t  kind;                    /* TRL_REC_MOVE / TRL_REC_GAME_END */
    int8_t   winner;                  /* GAME_END: Game.winner (game.py:217-225), -1 = draw */
    uint16_t chosen;                  /* MOVE: the move played */
    uint8_t  saved;                   /* MOVE: save_move or save_all (ai.py:1613) */
    uint8_t  random_move;             /* MOVE: a random starting move (ai.py:1588-1608): never saved, visits are zero */
    trl_game_state state;             /* MOVE: the position searched (game_to_X input, ai.py:1617); GAME_END: final */
    uint16_t moves[TRL_MAX_LEGAL];    /* root children: encoded moves ... */
    uint16_t visits[TRL_MAX_LEGAL];   /* ... and visit counts (search_statistics input, ai.py:1330) */
} trl_record;                         /* 2392 bytes (state is 8-byte aligned at offset 24) */

typedef struct trl_mcts_stats {
    uint64_t ticks;                   /* select/expand pairs executed */
    uint64_t simulations;             /* simulations executed over all slots */
    uint64_t moves;                   /* searches completed = moves played */
    uint64_t games;                   /* games finished */
    uint64_t records_dropped;         /* records lost because the device buffer was full */
    uint32_t max_nodes_used;          /* high-water mark of tree nodes in any slot */
    uint32_t overflow;                /* a slot exceeded max_nodes (its search was cut short) */
    uint64_t boards_reused;           /* leaf boards whose tower output was taken from the search's board table instead of
                                       * the tower: the same board reached under another reply of the opponent */
} trl_mcts_stats;

/* attach search state to an engine whose games are set up (trl_env_reset / trl_env_set_state) */
int  trl_mcts_create(trl_engine *h, const trl_mcts_config *cfg);
/* tick part A.  dev_grids: float[2][n_games][40][10] (side to move first, then opponent --
 * simplify_grid, ai.py:1364-1377), dev_vec: float[n_games][TRL_VEC_FLOATS]; both device pointers. */
int  trl_mcts_select(trl_engine *h, float *dev_grids, float *dev_vec);
/* Tower-output reuse.  A move changes only the mover's board (game.py:66-118: lock, clear, own garbage), so
 * the board of the side to move at a leaf is the board it had in the parent position and its tower output
 * (process_grid, architectures.py:121-127) can be reused bit for bit instead of recomputed.
 *   trl_mcts_select_need: as trl_mcts_select, and dev_need uint8[2][n_games] says which of the boards in
 *                         dev_grids need the tower ([0] side to move: only at a search's root; [1] the player
 *                         who just moved: every non-terminal leaf);
 *   trl_mcts_zcache:      dev_z float[2][n_games][400] holds the tower outputs of the flagged boards; the call
 *                         fills in the reused ones from the parent's cache entry and stores both outputs in the
 *                         leaf's entry, so that afterwards dev_z is complete for the heads. */
int  trl_mcts_select_need(trl_engine *h, float *dev_grids, float *dev_vec, uint8_t *dev_need);
/* on != 0: from now on trl_mcts_select_need leaves the leaf's legal-move search (get_move_matrix, 78 % of the select
 * kernel) to a second kernel that the library runs on a stream of its own BESIDE the caller's tower -- the tower needs the
 * leaf's boards only -- and the next trl_mcts_expand* call waits for it.  Inside trl_graph_begin / _end the fork and the
 * join become edges of the captured graph.  Results are the same bit for bit. */
int  trl_mcts_fork_legal(trl_engine *h, int on);
/* floats of tower output per board the cache holds (default 400 = AlphaSame's process_grid output; the BaseResNet
 * family caches its (own_kernels + 1) x 400 raw projections).  Call before the first tick: the cache is reallocated. */
int  trl_mcts_zcache_dim(trl_engine *h, int floats_per_board);
int  trl_mcts_zcache(trl_engine *h, float *dev_z);
/* tick part B.  dev_values: float[n_games] (sigmoid value head), dev_policy: float[n_games][11583]
 * softmaxed policy, both as evaluate() returns them (ai.py:1222-1319). */
int  trl_mcts_expand(trl_engine *h, const float *dev_values, const float *dev_policy);
/* tick part B without the 11 583-wide policy row: dev_feats float[n_games][feat_dim] is the input of
 * the policy head (521 for AlphaSame, architectures.py:101-111), dev_head_weight float[11583][feat_dim]
 * and dev_head_bias float[11583] the head (policy_head.weight / .bias).  Only the logits of the legal
 * moves are formed and p_i = exp(l_i - max over the legal moves) replaces the full softmax: the softmax
 * denominator cancels in the root temperature (ai.py:428-434) and in the normalisation over the legal
 * moves (ai.py:436-443), so the priors are the same up to float32 rounding (~1e-7 relative). */
int  trl_mcts_expand_head(trl_engine *h, const float *dev_values, const float *dev_feats, int feat_dim,
                          const float *dev_head_weight, const float *dev_head_bias);
/* tick part B for AlphaSame with every head inside the kernel: dev_z float[2][n_games][400] = tower outputs of
 * the boards trl_mcts_select_need flagged (the call completes them from the tower-output cache exactly as
 * trl_mcts_zcache does), dev_vec = the float[n_games][105] of trl_mcts_select.  The kernel forms the 521-vector
 * of architectures.py:132-140, the opponent-side dense layer, the value head and the legal moves' logits.
 * Weights: eval-mode BatchNorm1d folded into the preceding Linear (nets.AlphaSame.prepare()). */
typedef struct trl_alphasame_heads {
    const float *oside_w, *oside_b;       /* [16][400], [16]      osidedense.0 + .1 folded            */
    const float *value0_w, *value0_b;     /* [16][521], [16]      value_head.0 + .1 folded            */
    const float *value3_w;                /* [16]                 value_head.3.weight                  */
    const float *policy_w, *policy_b;     /* [11583][521], [11583] policy_head                         */
    float value3_b;                       /* value_head.3.bias                                          */
    int32_t use_tanh;                     /* output activation of the value head: 0 sigmoid, 1 tanh     */
} trl_alphasame_heads;
int  trl_mcts_expand_alphasame(trl_engine *h, const float *dev_z, const float *dev_vec, const trl_alphasame_heads *w);
/* the same followed by trl_mcts_select_need of the next simulation, in one launch: a slot goes on to its next selection
 * without waiting for the slowest expand of the batch.  dev_vec is read (the evaluated leaf's features), then overwritten
 * with the next leaf's; dev_grids / dev_need as in trl_mcts_select_need.  A tick is then {tower, this call}, after one
 * trl_mcts_select_need for the very first simulation.                                                               */
int  trl_mcts_expand_alphasame_select(trl_engine *h, const float *dev_z, float *dev_vec, const trl_alphasame_heads *w,
                                      float *dev_grids, uint8_t *dev_need);
/* every slot starts a new search at its next selection (the state after trl_mcts_create).  A caller that replaces the
 * game states (trl_env_set_state / trl_env_reset / trl_env_step) while a selection made by the call above is pending
 * discards it with this call and selects again (trl_mcts_select_need).                                              */
int  trl_mcts_restart(trl_engine *h);
/* The legal-move search of the leaf (get_move_matrix + get_move_list, ai.py:414-417) as part of the TOWER launch of the
 * tick instead of the selection: it needs the leaf's position only, exactly like the tower, so its workgroups run in the
 * registers the tower's boards leave free and the few searches that take four times as long as the others (T pieces
 * whose T-spin sub-plane depends on the walk's order) no longer set the length of the search kernel's launch.
 * on = 1: trl_mcts_select_need and trl_mcts_expand_alphasame_select (called with need flags) leave the search out; the
 * next call for this engine must be trl_mcts_tower16_legal, which runs the split 16-filter tower (the contract of
 * trl_alphasame_trunk16_split, on the engine's stream) with the search's workgroups in front.  Any other order fails
 * with TRL_E_STATE.  'convolutional' generator only; not together with trl_mcts_fork_legal. */
int  trl_mcts_legal_in_tower(trl_engine *h, int on);
int  trl_mcts_tower16_legal(trl_engine *h, const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                            float *dev_out, const uint8_t *dev_need);
/* the same for the slots with dev_slot_mask[slot] == want only (uint8 per slot): matches between two networks run one
 * launch per network (battle_networks, ai.py:1975-2052: a player's searches use that player's network); `last` = 1
 * on the launch that completes the tick. */
int  trl_mcts_expand_alphasame_masked(trl_engine *h, const float *dev_z, const float *dev_vec, const trl_alphasame_heads *w,
                                      const uint8_t *dev_slot_mask, int want, int last);
/* copy out and clear the pending records (host buffer of `cap` records); returns the count in *n_out */
int  trl_mcts_drain_records(trl_engine *h, trl_record *host_out, int cap, int *n_out);
/* the same into a DEVICE buffer of `cap` records (the gather of the records to rank 0 then runs GPU -> GPU over
 * RCCL / xGMI without touching the sender's host); the copy is queued on the engine's stream, the count is returned */
int  trl_mcts_drain_records_dev(trl_engine *h, trl_record *dev_out, int cap, int *n_out);
/* the same without a host round trip: the pending records are APPENDED to dev_out behind the dev_head[0] records already
 * there (room for `cap` in all; what does not fit is counted in records_dropped), every record's slot raised by
 * slot_offset (a slot group's first global slot), and the pending count cleared -- all queued on the engine's stream, the
 * call returns at once.  dev_head: uint64[9] on the device, zeroed by the caller before the first call;
 * [0] = records in dev_out, [1..8] = the search counters as of this drain: simulations, moves, games, records_dropped,
 * max_nodes_used, overflow, ticks, boards_reused.  A job that ticks in chunks queues the next chunk behind this call and
 * reads dev_head / dev_out of the previous one on another stream: the device never waits for the host (the reference's
 * loop appends a move's record to the game's list as it is played, ai.py:1613-1660; this is that list leaving the
 * search's buffer). */
int  trl_mcts_drain_records_async(trl_engine *h, trl_record *dev_out, int cap, uint64_t *dev_head, uint32_t slot_offset);
int  trl_mcts_get_stats(trl_engine *h, trl_mcts_stats *host_out);
/* leaf positions of the current tick (after trl_mcts_select), n_games canonical states; and per
 * slot {simulation index, simulations of this search, leaf kind 0 expand / 1 eval only / 2 terminal,
 * legal count} -- what a host-side evaluator or a test needs */
int  trl_mcts_leaf_states(trl_engine *h, trl_game_state *host_out, int32_t *host_info /* [n_games][4] */);
/* root children of every slot's current tree: counts[n], moves / visits [n][TRL_MAX_LEGAL], q / p as fp64 */
int  trl_mcts_root_children(trl_engine *h, int32_t *counts, uint16_t *moves, int32_t *visits, double *q, double *p);

/* ==== HIP graph of a search tick ==================================================================
 * Everything issued on the engine's stream between trl_graph_begin and trl_graph_end (trl_mcts_select*,
 * the tower, trl_mcts_expand* -- device pointers only, no host copies) is captured into one hipGraph;
 * trl_graph_launch replays it n_times on the same stream.  Replaces the reference's Python loop over
 * MAX_ITER iterations (ai.py:330-338) as the unit of host work.
 * Optionally a second capture of the same calls issued `ticks` (>= 2) times in a row, ended by
 * trl_graph_end_many (after trl_graph_end): trl_graph_launch then replays that graph n_times / ticks
 * times and the single tick for the remainder -- the stream does not idle between the ticks of one graph. */
int  trl_graph_begin(trl_engine *h);
int  trl_graph_end(trl_engine *h);
int  trl_graph_end_many(trl_engine *h, int ticks);
int  trl_graph_launch(trl_engine *h, int n_times);

/* ==== net: residual tower of AlphaSame on the device (architectures.py:27-57, 119-127) ===========
 * process_grid for filters = 16, kernels = 1: conv5x5 -> n_blocks residual blocks -> BN, ReLU ->
 * conv1x1 -> BN, ReLU -> flatten, one workgroup per board, fp32 MFMA, activations resident in LDS.
 * dev_boards float[n_boards][400], dev_out float[n_boards][400]; dev_packed = the weights with the
 * eval-mode batch norms folded, in the layout documented in csrc/trl_net.hip (built by
 * nets.AlphaSame.prepare()): conv1 [16][25]; per block s1[16] t1[16] Wa[9][4][64] ba[16] Wb[9][4][64];
 * tail s[16] t[16] wk[16] bk.  stream: hipStream_t (NULL = default stream).                        */
int  trl_alphasame_trunk16(const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                           float *dev_out, void *stream);
/* same, skipping the boards whose dev_need[board] is 0 (their dev_out rows are left untouched) */
int  trl_alphasame_trunk16_need(const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                                float *dev_out, const uint8_t *dev_need, void *stream);
/* The same tower at the f16 matrix rate with fp32 accuracy: every fp32 operand (activations in the kernel, weights on the
 * host) is split into two halves, a = hi + lo / 2^11, and a product becomes three v_mfma_f32_16x16x32_f16 (hi hi, lo hi,
 * hi lo) accumulated in fp32 -- 15 matrix instructions of 16 cycles per (16-position tile, conv) instead of 36 of 32.
 * Logits within 3e-7 of the reference's, as the fp32 MFMA tower (csrc/trl_net.hip k_trunk16s).  dev_packed: conv1
 * [16][25] fp32; per block s1[16] t1[16] Wa[5][2][64][8 halves] ba[16] Wb[...]; tail as above
 * (nets.AlphaSame._pack_tower_split).  dev_need may be NULL.                                                         */
int  trl_alphasame_trunk16_split(const float *dev_boards, int n_boards, const float *dev_packed, int n_blocks,
                                 float *dev_out, const uint8_t *dev_need, void *stream);
/* The 32- and 64-filter towers in the same way (csrc/trl_net.hip k_trunk_wide_s): resnet = 0: AlphaSame, contract of
 * trl_alphasame_trunk (kout ignored); resnet != 0: the BaseResNet family, contract of trl_resnet_trunk.  dev_packed as for
 * those, with every 3x3 conv as [K-step = kg * 9 + tap][hi | lo][N-tile][64 lanes][8 halves] (nets._pack_conv3x3_split_wide). */
int  trl_trunk_wide_split(const float *dev_boards, int n_boards, const float *dev_packed, int filters, int n_blocks, int resnet,
                          int kout, float *dev_out, const uint8_t *dev_need, void *stream);
/* the same tower for filters in {16, 32, 64} (BASELINE config C5 is 20 blocks x 64 filters): F = filters,
 * G = F/16 channel groups and N-tiles.  dev_packed: conv1 [F][25]; per block s1[F] t1[F] Wa ba[F] Wb with a 3x3
 * conv packed as [G channel groups][9 taps][4 j][G N-tiles][64 lanes] = W[16 ntile + lane%16][16 cg + 4 (lane/16) + j][tap]
 * (the B operands of v_mfma_f32_16x16x4_f32; at F = 16 this is the trunk16 layout); tail s[F] t[F] wk[F] bk, then
 * >= 16 KiB of padding (the weight prefetch of the last conv runs one step ahead).  dev_need may be NULL. */
int  trl_alphasame_trunk(const float *dev_boards, int n_boards, const float *dev_packed, int filters, int n_blocks,
                         float *dev_out, const uint8_t *dev_need, void *stream);

/* Tower of the BaseResNet family -- the reference's current default nets (architectures.py:159-170 ResidualBlock,
 * 173-222 BaseResNet / 308-353 AuxBaseResNet process_grid): stem conv3x3 + BN + ReLU, n_blocks post-activation
 * residual blocks, filters in {32, 64}; followed by kout (1..8) raw 1x1 projections of the final feature map
 * (own_collapse / opp_collapse before bias, conditioning and ReLU, which the caller applies).
 * dev_out float[n_boards][kout][400].  dev_packed (every BN folded into its conv): stem W[F][9] b[F]; per block
 * b1[F] W1 b2[F] W2 (3x3 convs packed as for trl_alphasame_trunk); tail Wk[kout][F]; >= 16 KiB of padding. */
int  trl_resnet_trunk(const float *dev_boards, int n_boards, const float *dev_packed, int filters, int n_blocks, int kout,
                      float *dev_out, const uint8_t *dev_need, void *stream);

/* Range guard of the split towers (trl_alphasame_trunk16_split, trl_trunk_wide_split).  They split every activation into
 * two f16 halves; an activation of 65 504 or more has no finite hi half and the board's output would be built on inf / NaN
 * (the folded conv WEIGHTS are range-checked on the host when they are packed, activations can only be checked here).
 * *count_out = 0 if that has not happened on HIP device `device` since the last reset, else the number of wavefronts (4
 * or 8 per board) that saw it; reset != 0 clears the count.
 * The caller synchronises the stream(s) its towers ran on first.  A caller that sees a count > 0 runs those boards again
 * on the fp32 towers (trl_alphasame_trunk / trl_resnet_trunk), which have no such limit -- nets.AlphaSame.trunk() does.    */
int  trl_tower_overflow(int device, int reset, uint32_t *count_out);

#ifdef __cplusplus
}
#endif
#endif /* TETRIS_ENGINE_H */
