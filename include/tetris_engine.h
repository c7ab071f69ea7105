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

#ifdef __cplusplus
}
#endif
#endif /* TETRIS_ENGINE_H */
