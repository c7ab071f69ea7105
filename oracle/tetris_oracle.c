/*
 * tetris_oracle.c -- CPU restatement of the reference environment (test infrastructure,
 * see tetris_oracle.h).  Every function cites the reference file:line it follows.
 * Plain C99, no dependencies.
 */
#include "tetris_oracle.h"
#include <string.h>
#include <stdlib.h>
#include <math.h>

/* ------------------------------------------------------------------------------------------
 * tables (const.py)
 * ---------------------------------------------------------------------------------------- */
enum { PZ = 0, PL = 1, PO = 2, PS = 3, PI = 4, PJ = 5, PT = 6 };   /* MINOS = "ZLOSIJT", const.py:68 */

/* mino_coords_dict[type][rot][mino] = {col,row}, const.py:238-281 */
static const int8_t MINO[7][4][4][2] = {
    /* Z */ {{{0,0},{1,0},{1,1},{2,1}}, {{1,1},{1,2},{2,0},{2,1}}, {{0,1},{1,1},{1,2},{2,2}}, {{0,1},{0,2},{1,0},{1,1}}},
    /* L */ {{{0,1},{1,1},{2,0},{2,1}}, {{1,0},{1,1},{1,2},{2,2}}, {{0,1},{0,2},{1,1},{2,1}}, {{0,0},{1,0},{1,1},{1,2}}},
    /* O */ {{{0,0},{0,1},{1,0},{1,1}}, {{0,0},{0,1},{1,0},{1,1}}, {{0,0},{0,1},{1,0},{1,1}}, {{0,0},{0,1},{1,0},{1,1}}},
    /* S */ {{{0,1},{1,0},{1,1},{2,0}}, {{1,0},{1,1},{2,1},{2,2}}, {{0,2},{1,1},{1,2},{2,1}}, {{0,0},{0,1},{1,1},{1,2}}},
    /* I */ {{{0,1},{1,1},{2,1},{3,1}}, {{2,0},{2,1},{2,2},{2,3}}, {{0,2},{1,2},{2,2},{3,2}}, {{1,0},{1,1},{1,2},{1,3}}},
    /* J */ {{{0,0},{0,1},{1,1},{2,1}}, {{1,0},{1,1},{1,2},{2,0}}, {{0,1},{1,1},{2,1},{2,2}}, {{0,2},{1,0},{1,1},{1,2}}},
    /* T */ {{{0,1},{1,0},{1,1},{2,1}}, {{1,0},{1,1},{1,2},{2,1}}, {{0,1},{1,1},{1,2},{2,1}}, {{0,1},{1,0},{1,1},{1,2}}},
};

/* len(piece_dict[type]), const.py:138-174 */
static const int BOX[7] = {3, 3, 2, 3, 4, 3, 3};

/* wallkicks[from][to] -> list of {x,y}; const.py:191-212.  n = 0 means "no entry". */
typedef struct { int n; int8_t k[6][2]; } kicklist;
static const kicklist KICKS[4][4] = {
  /* from 0 */ { {0,{{0}}},
                 {5,{{0,0},{-1,0},{-1,1},{0,-2},{-1,-2}}},
                 {6,{{0,0},{0,1},{1,1},{-1,1},{1,0},{-1,0}}},
                 {5,{{0,0},{1,0},{1,1},{0,-2},{1,-2}}} },
  /* from 1 */ { {5,{{0,0},{1,0},{1,-1},{0,2},{1,2}}},
                 {0,{{0}}},
                 {5,{{0,0},{1,0},{1,-1},{0,2},{1,2}}},
                 {6,{{0,0},{1,0},{1,2},{1,1},{0,2},{0,1}}} },
  /* from 2 */ { {6,{{0,0},{0,-1},{-1,-1},{1,-1},{-1,0},{1,0}}},
                 {5,{{0,0},{-1,0},{-1,1},{0,-2},{-1,-2}}},
                 {0,{{0}}},
                 {5,{{0,0},{1,0},{1,1},{0,-2},{1,-2}}} },
  /* from 3 */ { {5,{{0,0},{-1,0},{-1,-1},{0,2},{-1,2}}},
                 {6,{{0,0},{-1,0},{-1,2},{-1,1},{0,2},{0,1}}},
                 {5,{{0,0},{-1,0},{-1,-1},{0,2},{-1,2}}},
                 {0,{{0}}} },
};
/* i_wallkicks, const.py:214-235 */
static const kicklist IKICKS[4][4] = {
  /* from 0 */ { {0,{{0}}},
                 {5,{{0,0},{-2,0},{1,0},{-2,-1},{1,2}}},
                 {2,{{0,0},{0,1}}},
                 {5,{{0,0},{-1,0},{2,0},{-1,2},{2,-1}}} },
  /* from 1 */ { {5,{{0,0},{2,0},{-1,0},{2,1},{-1,-2}}},
                 {0,{{0}}},
                 {5,{{0,0},{-1,0},{2,0},{-1,2},{2,-1}}},
                 {2,{{0,0},{1,0}}} },
  /* from 2 */ { {2,{{0,0},{0,-1}}},
                 {5,{{0,0},{1,0},{-2,0},{1,-2},{-2,1}}},
                 {0,{{0}}},
                 {5,{{0,0},{2,0},{-1,0},{2,1},{-1,-2}}} },
  /* from 3 */ { {5,{{0,0},{1,0},{-2,0},{1,-2},{-2,1}}},
                 {2,{{0,0},{-1,0}}},
                 {5,{{0,0},{-2,0},{1,0},{-2,-1},{1,2}}},
                 {0,{{0}}} },
};

/* policy_index_to_piece, const.py:82-118 */
static const int8_t PLANE_PIECE[TRL_PLANES][3] = {
    {PO,0,0},
    {PZ,0,0},{PZ,1,0},
    {PS,0,0},{PS,1,0},
    {PI,0,0},{PI,1,0},
    {PL,0,0},{PL,1,0},{PL,2,0},{PL,3,0},
    {PJ,0,0},{PJ,1,0},{PJ,2,0},{PJ,3,0},
    {PT,0,0},{PT,1,0},{PT,2,0},{PT,3,0},
    {PT,0,1},{PT,1,1},{PT,2,1},{PT,3,1},
    {PT,0,2},{PT,1,2},{PT,2,2},{PT,3,2},
};
/* policy_pieces / policy_piece_to_index, const.py:71-80,125-136 */
static const int N_ROT[7]      = {2, 4, 1, 2, 2, 4, 4};
static const int PLANE_BASE[7] = {1, 7, 0, 3, 5, 11, 15};
static int policy_index(int type, int rot_index, int tspin) {
    return PLANE_BASE[type] + rot_index + (type == PT ? 4 * tspin : 0);
}

/* ------------------------------------------------------------------------------------------
 * rng (include/tetris_engine.h)
 * ---------------------------------------------------------------------------------------- */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
uint32_t to_rng_u32(uint64_t seed, uint32_t stream, uint32_t purpose, uint64_t ctr) {
    uint64_t k = mix64(seed ^ ((uint64_t)stream * 0xD1342543DE82EF95ull));
    return (uint32_t)(mix64(k ^ ((uint64_t)purpose << 56) ^ ctr) >> 32);
}
static int rng_below(const to_game *g, uint32_t purpose, uint64_t ctr, uint32_t n) {
    return (int)(((uint64_t)to_rng_u32(g->seed, g->stream, purpose, ctr) * n) >> 32);
}

/* ------------------------------------------------------------------------------------------
 * board.py
 * ---------------------------------------------------------------------------------------- */
typedef struct { int x, y, rot, rot_flag, last_kick_flag; } pose;   /* piece_location.py:3-14 */

/* Board.is_valid_position for the 4 minos of (type, x, y, rot) -- board.py:23-30, piece.py:50-55 */
static int valid_pose(const to_player *pl, int type, int x, int y, int rot) {
    for (int m = 0; m < 4; m++) {
        int col = x + MINO[type][rot][m][0], row = y + MINO[type][rot][m][1];
        if (row < 0 || row > TRL_ROWS - 1 || col < 0 || col > TRL_COLS - 1) return 0;
        if (pl->grid[row][col] != 0) return 0;
    }
    return 1;
}
static int valid_cell(const to_player *pl, int col, int row) {
    if (row < 0 || row > TRL_ROWS - 1 || col < 0 || col > TRL_COLS - 1) return 0;
    return pl->grid[row][col] == 0;
}

/* Board.create_garbage -- board.py:12-21 */
static void create_garbage(to_player *pl, const int *cols, int n) {
    if (n > TRL_ROWS) n = TRL_ROWS;   /* the reference raises for n > ROWS; callers flag overflow */
    memmove(&pl->grid[0][0], &pl->grid[n][0], (size_t)(TRL_ROWS - n) * TRL_COLS);
    for (int i = 0; i < n; i++) {
        memset(pl->grid[TRL_ROWS - n + i], 1, TRL_COLS);
        pl->grid[TRL_ROWS - n + i][cols[i]] = 0;
    }
}

/* ------------------------------------------------------------------------------------------
 * stats.py
 * ---------------------------------------------------------------------------------------- */
/* Stats.update_b2b_level -- stats.py:30-43 */
static void update_b2b_level(to_player *p) {
    static const int chart[9][2] = {{-1,0},{1,1},{3,2},{8,3},{24,4},{67,5},{185,6},{504,7},{1370,8}};
    for (int i = 0; i < 9; i++)
        if (p->b2b >= chart[i][0] && p->b2b_level < chart[i][1]) p->b2b_level = chart[i][1];
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* Stats.get_attack -- stats.py:45-157 (floats as in the reference: all values are exact quarters) */
int to_get_attack(to_player *p, int ruleset, int rows_cleared, int is_tspin, int is_mini, int is_all_clear) {
    int attack = 0;
    is_tspin = !!is_tspin; is_mini = !!is_mini;
    if (ruleset == TRL_RULESET_S1) {                                   /* stats.py:49-86 */
        if (rows_cleared == 0) {
            p->combo = 0;
        } else {
            int is_b2b = (rows_cleared == 4 || is_tspin);
            if (is_b2b) p->b2b += 1; else p->b2b = -1;
            update_b2b_level(p);
            if (rows_cleared == 1) {
                if (!is_tspin) {
                    p->b2b = -1;
                    attack += (int)floor(0.5 + 0.25 * p->combo);
                } else if (is_mini) {
                    if (p->b2b <= 0 && p->b2b_level <= 0) attack += (int)floor(0.5 + 0.25 * p->combo);
                    else attack += (int)floor(p->b2b_level * (1 + 0.25 * p->combo));
                } else {
                    attack += (int)floor((2 + p->b2b_level) * (1 + 0.25 * p->combo));
                }
            } else {
                attack += (int)floor((1 + 0.25 * p->combo) *
                          (2 * rows_cleared * is_tspin * (-3.0 / 4 * is_mini + 1)
                           + (double)(1 << (rows_cleared - 2)) * (1 - is_tspin)
                           + p->b2b_level * (is_tspin || rows_cleared == 4)));
            }
            p->combo += 1;
            if (is_all_clear) attack += 10;
        }
    } else {                                                            /* 's2', stats.py:88-133 */
        if (rows_cleared == 0) {
            p->combo = 0;
        } else {
            int is_b2b = (is_tspin || is_mini || rows_cleared == 4);
            if (is_b2b) {
                p->b2b += 1;
            } else if (is_all_clear) {
                attack += 5;
                p->b2b += 1;
            } else {
                if (p->b2b >= 4) attack += p->b2b;     /* surge */
                p->b2b = -1;
            }
            update_b2b_level(p);
            int lvl = imin(imax(p->b2b_level, 0), 1);
            if (rows_cleared == 1) {
                if (!is_tspin) {
                    attack += (int)floor(0.5 + 0.25 * p->combo);
                } else if (is_mini) {
                    if (p->b2b <= 0 && p->b2b_level <= 0) attack += (int)floor(0.5 + 0.25 * p->combo);
                    else attack += (int)floor(lvl * (1 + 0.25 * p->combo));
                } else {
                    attack += (int)floor(2 + lvl * (1 + 0.25 * p->combo));
                }
            } else {
                attack += (int)floor((1 + 0.25 * p->combo) *
                          (2 * rows_cleared * is_tspin * (-3.0 / 4 * is_mini + 1)
                           + (double)(1 << (rows_cleared - 2)) * (1 - is_tspin)
                           + lvl * is_b2b));
            }
            p->combo += 1;
        }
    }
    p->lines_sent += attack;                                            /* stats.py:135-136 */
    p->lines_cleared += rows_cleared;
    return attack;
}

/* ------------------------------------------------------------------------------------------
 * player.py
 * ---------------------------------------------------------------------------------------- */
static int spawn_x(int type) { return 3 + (type == PO ? 1 : 0); }     /* piece.py:17-21 */
#define SPAWN_Y (TRL_ROWS - TRL_SPAWN_ROW)

/* Player.create_piece -- player.py:37-44 */
static void create_piece(to_player *pl, int type) {
    if (valid_pose(pl, type, spawn_x(type), SPAWN_Y, 0)) pl->piece = type;
    else pl->game_over = 1;
}
/* Player.create_next_piece -- player.py:29-35 */
static void create_next_piece(to_player *pl) {
    if (pl->queue_len > 0) {
        int next = pl->queue[0];
        memmove(pl->queue, pl->queue + 1, sizeof(int) * (size_t)(pl->queue_len - 1));
        pl->queue_len--;
        create_piece(pl, next);
    } else if (pl->held != TRL_NO_PIECE) {
        int next = pl->held;
        pl->held = TRL_NO_PIECE;
        create_piece(pl, next);
    }
}
/* Player.hold_piece -- player.py:190-201 (the caller guarantees piece or held is set) */
static void hold_piece(to_player *pl) {
    if (pl->held == TRL_NO_PIECE) {
        pl->held = pl->piece;
        pl->piece = TRL_NO_PIECE;
        create_next_piece(pl);
    } else {
        int temp = pl->held;
        pl->held = pl->piece;            /* None stays None */
        create_piece(pl, temp);          /* on failure the old active piece stays in place */
    }
}

/* Player.place_piece -- player.py:109-188.  loc = pose set by Game.move_piece. */
static int place_piece(to_game *g, to_player *pl, pose loc) {
    int type = pl->piece;
    /* ghost_y, player.py:46-59 */
    int delta = 1;
    while (valid_pose(pl, type, loc.x, loc.y + delta, loc.rot)) delta++;
    int place_y = loc.y + delta - 1;

    if (loc.y != place_y) { loc.rot_flag = 0; loc.last_kick_flag = 0; }     /* :115-118 */

    int is_tspin = 0, is_mini = 0, is_all_clear = 0;
    if (type == PT && loc.rot_flag) {                                        /* :127-140 */
        static const int corners[4][2] = {{0,0},{2,0},{2,2},{0,2}};
        int filled[4], sum = 0;
        for (int i = 0; i < 4; i++) {
            filled[i] = !valid_cell(pl, corners[i][0] + loc.x, corners[i][1] + place_y);
            sum += filled[i];
        }
        if (sum >= 3) is_tspin = 1;
        if (!(filled[loc.rot] && filled[(loc.rot + 1) % 4]) && !loc.last_kick_flag) is_mini = 1;
    }
    if (g->ruleset == TRL_RULESET_S2 && !is_tspin) {                         /* :143-151, coords at loc.y */
        static const int offs[4][2] = {{-1,0},{1,0},{0,-1},{0,1}};
        int mobile = 0;
        for (int i = 0; i < 4; i++)
            if (valid_pose(pl, type, loc.x + offs[i][0], loc.y + offs[i][1], loc.rot)) { mobile = 1; break; }
        if (!mobile) is_mini = 1;
    }

    int rows[4], n_rows = 0;                                                 /* :153-157 */
    for (int m = 0; m < 4; m++) {
        int col = loc.x + MINO[type][loc.rot][m][0], row = place_y + MINO[type][loc.rot][m][1];
        pl->grid[row][col] = (uint8_t)(2 + type);
        int seen = 0;
        for (int i = 0; i < n_rows; i++) if (rows[i] == row) seen = 1;
        if (!seen) rows[n_rows++] = row;
    }
    pl->piece = TRL_NO_PIECE;

    int cleared[4], n_cleared = 0;                                           /* :161-168 */
    for (int i = 0; i < n_rows; i++) {
        int full = 1;
        for (int c = 0; c < TRL_COLS; c++) if (pl->grid[rows[i]][c] == 0) full = 0;
        if (full) cleared[n_cleared++] = rows[i];
    }
    if (n_cleared > 0) {                                                     /* :170-175 delete + zero rows on top */
        uint8_t ng[TRL_ROWS][TRL_COLS];
        memset(ng, 0, sizeof ng);
        int dst = TRL_ROWS - 1;
        for (int r = TRL_ROWS - 1; r >= 0; r--) {
            int is_c = 0;
            for (int i = 0; i < n_cleared; i++) if (cleared[i] == r) is_c = 1;
            if (!is_c) { memcpy(ng[dst], pl->grid[r], TRL_COLS); dst--; }
        }
        memcpy(pl->grid, ng, sizeof ng);
        is_all_clear = 1;                                                    /* :177-178 */
        for (int r = n_cleared; r < TRL_ROWS; r++)
            for (int c = 0; c < TRL_COLS; c++) if (pl->grid[r][c] != 0) is_all_clear = 0;
    }

    int attack = to_get_attack(pl, g->ruleset, n_cleared, is_tspin, is_mini, is_all_clear);   /* :180 */
    pl->pieces += 1;
    if (attack > 0) {                                                        /* :183-186 */
        int column = rng_below(g, TRL_RNG_GARBAGE, g->garb_ctr++, 10);       /* random.randint(0, 9) */
        for (int i = 0; i < attack; i++) {
            if (pl->garbage_send_len < TO_MAX_GARBAGE) pl->garbage_send[pl->garbage_send_len++] = column;
            else g->overflow = 1;
        }
    }
    return n_cleared;
}

/* ------------------------------------------------------------------------------------------
 * piece_queue.py / game.py
 * ---------------------------------------------------------------------------------------- */
/* Queue.generate_bag -- piece_queue.py:17-21, with the engine's Fisher-Yates in place of random.shuffle */
static void generate_bag(to_game *g, int bag[7]) {
    for (int i = 0; i < 7; i++) bag[i] = i;
    uint64_t k = g->bag_ctr++;
    for (int i = 6; i >= 1; i--) {
        int j = rng_below(g, TRL_RNG_BAG, 8 * k + (uint64_t)i, (uint32_t)(i + 1));
        int t = bag[i]; bag[i] = bag[j]; bag[j] = t;
    }
}
/* Game.add_bag_to_all -- game.py:34-38 */
static void add_bag_to_all(to_game *g) {
    for (int i = 0; i < 2; i++) {
        int bag[7];
        generate_bag(g, bag);
        for (int k = 0; k < 7; k++) {
            if (g->p[i].queue_len < TO_MAX_QUEUE) g->p[i].queue[g->p[i].queue_len++] = bag[k];
            else g->overflow = 1;
        }
    }
}

void to_game_setup(to_game *g, int ruleset, uint64_t seed, uint32_t stream) {   /* game.py:8-32, player.py:12-27 */
    memset(g, 0, sizeof *g);
    g->ruleset = ruleset; g->seed = seed; g->stream = stream;
    for (int i = 0; i < 2; i++) {
        g->p[i].piece = TRL_NO_PIECE; g->p[i].held = TRL_NO_PIECE;
        g->p[i].b2b = -1;
    }
    add_bag_to_all(g);
    for (int i = 0; i < 2; i++) create_next_piece(&g->p[i]);
}

/* Game.check_garbage -- game.py:100-117 */
static void check_garbage(to_game *g, int line_clear) {
    to_player *a = &g->p[g->turn], *o = &g->p[1 - g->turn];
    int cancel = imin(a->garbage_send_len, a->garbage_recv_len);             /* pop(0) pairs */
    if (cancel > 0) {
        memmove(a->garbage_send, a->garbage_send + cancel, sizeof(int) * (size_t)(a->garbage_send_len - cancel));
        memmove(a->garbage_recv, a->garbage_recv + cancel, sizeof(int) * (size_t)(a->garbage_recv_len - cancel));
        a->garbage_send_len -= cancel; a->garbage_recv_len -= cancel;
    }
    if (a->garbage_recv_len > 0 && !line_clear) {                            /* spawn_garbage, player.py:203-205 */
        if (a->garbage_recv_len > TRL_ROWS) g->overflow = 1;
        create_garbage(a, a->garbage_recv, a->garbage_recv_len);
        a->garbage_recv_len = 0;
    }
    if (a->garbage_send_len > 0) {
        for (int i = 0; i < a->garbage_send_len; i++) {
            if (o->garbage_recv_len < TO_MAX_GARBAGE) o->garbage_recv[o->garbage_recv_len++] = a->garbage_send[i];
            else g->overflow = 1;
        }
        a->garbage_send_len = 0;
    }
}

void to_make_move(to_game *g, int plane, int col, int row, int add_bag) {
    to_player *pl = &g->p[g->turn];
    /* Game.move_piece -- game.py:40-64 */
    int type = PLANE_PIECE[plane][0], rot = PLANE_PIECE[plane][1], ts = PLANE_PIECE[plane][2];
    if (pl->piece == TRL_NO_PIECE) hold_piece(pl);
    else if (type != pl->piece) hold_piece(pl);
    pose loc = { col, row, rot, 0, 0 };
    if (ts == 1) loc.rot_flag = 1;
    else if (ts == 2) { loc.last_kick_flag = 1; loc.rot_flag = 1; }
    /* Game.place -- game.py:66-94 */
    if (!pl->game_over) {
        int rows_cleared = place_piece(g, pl, loc);
        check_garbage(g, rows_cleared > 0);
        create_next_piece(pl);
        if (add_bag && pl->queue_len < 5) add_bag_to_all(g);
        g->turn = 1 - g->turn;
        g->ply++;
    }
}

int to_is_terminal(const to_game *g) { return g->p[0].game_over || g->p[1].game_over; }
int to_no_move(const to_game *g) {
    const to_player *pl = &g->p[g->turn];
    return pl->piece == TRL_NO_PIECE && pl->held == TRL_NO_PIECE;
}
int to_winner(const to_game *g) {
    if (g->p[0].game_over) return 1;
    if (g->p[1].game_over) return 0;
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * move_generation.py
 * ---------------------------------------------------------------------------------------- */
#define MAP_W 14
#define MAP_H 44
#define MAX_PLACE 8192

typedef struct { int8_t x, y, rot, flag, last; } placement;

typedef struct {
    const to_player *board;      /* player.board (never written) */
    int type;                    /* self.piece.type */
    placement *place; int n_place;
} movegen;

/* MoveGenerator._get_highest_row -- move_generation.py:174-180 */
static int highest_row(const to_player *pl) {
    for (int r = 0; r < TRL_ROWS; r++)
        for (int c = 0; c < TRL_COLS; c++) if (pl->grid[r][c]) return r;
    return TRL_ROWS;
}

static void add_place(movegen *mg, int x, int y, int rot, int flag, int last) {
    if (mg->n_place < MAX_PLACE) {
        placement p = {(int8_t)x, (int8_t)y, (int8_t)rot, (int8_t)flag, (int8_t)last};
        mg->place[mg->n_place++] = p;
    }
}

/* MoveGenerator._build_validity_maps -- move_generation.py:490-528
 * valid[r][ym] bit xm: piece (rotation r) with origin (xm-2, ym-2) has all minos in bounds and empty */
static void build_validity_maps(const movegen *mg, uint16_t valid[4][MAP_H]) {
    for (int r = 0; r < 4; r++)
        for (int ym = 0; ym < MAP_H; ym++) {
            uint16_t bits = 0;
            for (int xm = 0; xm < MAP_W; xm++)
                if (valid_pose(mg->board, mg->type, xm - 2, ym - 2, r)) bits |= (uint16_t)(1u << xm);
            valid[r][ym] = bits;
        }
}

/* MoveGenerator._convolutional_algorithm -- move_generation.py:325-488 */
static void convolutional(movegen *mg, int start_x, int start_y, int check_rotations) {
    static const int x_off = 2, y_off = 2;
    uint16_t valid[4][MAP_H];
    uint16_t visited[4][MAP_H];
    build_validity_maps(mg, valid);
    memset(visited, 0, sizeof visited);

    int spawn_r = 0;
    int spawn_map_x = start_x + x_off, spawn_map_y = start_y + y_off;
    if (!(valid[spawn_r][spawn_map_y] & (1u << spawn_map_x))) return;        /* :349-351 topped out */

    /* exploration_queue: deque of (x, y, rot, rot_just_occurred, used_last_kick) */
    int qcap = 4096, qh = 0, qt = 0;
    placement *q = (placement *)malloc(sizeof(placement) * (size_t)qcap);
    q[qt++] = (placement){(int8_t)start_x, (int8_t)start_y, (int8_t)spawn_r, 0, 0};

    const kicklist (*table)[4] = (mg->type == PI) ? IKICKS : KICKS;          /* :358 */
    int is_T = (mg->type == PT);

    while (qh < qt) {
        placement it = q[qh++];
        int rot = it.rot;
        int map_x = it.x + x_off, map_y = it.y + y_off;
        unsigned map_bit = 1u << map_x;

        if (it.flag) {                                                       /* :388-394 */
            int below = map_y + 1;
            int stuck = (below >= MAP_H) || !(valid[rot][below] & map_bit);
            if (stuck) add_place(mg, it.x, it.y, rot, 1, it.last);
        }
        if (visited[rot][map_y] & map_bit) continue;                         /* :397-398 */
        if (!(valid[rot][map_y] & map_bit)) continue;                        /* :406-407 */

        unsigned reachable[MAP_H];
        memset(reachable, 0, sizeof reachable);
        reachable[map_y] = map_bit;

        for (int fy = map_y; fy < MAP_H; fy++) {                             /* :412-488 */
            unsigned r = reachable[fy];
            if (!r) continue;
            unsigned vr = valid[rot][fy];
            unsigned unvisited_vr = vr & ~(unsigned)visited[rot][fy];
            for (;;) {
                unsigned nr = r | ((r << 1) & unvisited_vr) | ((r >> 1) & unvisited_vr);
                if (nr == r) break;
                r = nr;
            }
            visited[rot][fy] |= (uint16_t)r;

            unsigned next_vr = (fy + 1 < MAP_H) ? valid[rot][fy + 1] : 0;
            unsigned blocked_down = r & ~next_vr;
            unsigned edges = blocked_down | (r & ~(vr << 1)) | (r & ~(vr >> 1));

            int ey = fy - y_off;
            for (int bit = 0; bit < 32; bit++) {                             /* LSB iteration, :433-481 */
                unsigned lsb = 1u << bit;
                if (!(edges & lsb)) continue;
                int edge_x = bit - x_off;
                if (blocked_down & lsb) add_place(mg, edge_x, ey, rot, 0, 0);
                if (!check_rotations) continue;
                for (int kick_dir = 1; kick_dir <= 3; kick_dir++) {
                    int new_rot = (rot + kick_dir) % 4;
                    const kicklist *kl = &table[rot][new_rot];
                    for (int ki = 0; ki < kl->n; ki++) {
                        int new_x = edge_x + kl->k[ki][0];
                        int new_y = ey - kl->k[ki][1];
                        int nmx = new_x + x_off, nmy = new_y + y_off;
                        unsigned new_bit = (nmx >= 0 && nmx < MAP_W) ? (1u << nmx) : 0;
                        if (new_bit && nmy >= 0 && nmy < MAP_H && (valid[new_rot][nmy] & new_bit)) {
                            int new_last = is_T && kick_dir != 2 && ki == kl->n - 1;
                            if (new_y < 0) break;                            /* :462-463 */
                            if (!(visited[new_rot][nmy] & new_bit)) {
                                if (qt == qcap) { qcap *= 2; q = (placement *)realloc(q, sizeof(placement) * (size_t)qcap); }
                                q[qt++] = (placement){(int8_t)new_x, (int8_t)new_y, (int8_t)new_rot, 1, (int8_t)new_last};
                            } else {
                                int below = nmy + 1;
                                if (below >= MAP_H || !(valid[new_rot][below] & new_bit))
                                    add_place(mg, new_x, new_y, new_rot, 1, new_last);
                            }
                            break;                                           /* first successful kick wins */
                        }
                    }
                }
            }
            if (fy + 1 < MAP_H) {
                unsigned nreach = r & next_vr & ~(unsigned)visited[rot][fy + 1];
                if (nreach) reachable[fy + 1] |= nreach;
            }
        }
    }
    free(q);
}

/* Player.try_wallkick -- player.py:77-102 */
static int try_wallkick(const to_player *pl, int type, pose *loc, int dir) {
    int initial = loc->rot, final = (initial + dir) % 4;
    const kicklist *kl = (type == PI) ? &IKICKS[initial][final] : &KICKS[initial][final];
    for (int ki = 0; ki < kl->n; ki++) {
        int nx = loc->x + kl->k[ki][0], ny = loc->y - kl->k[ki][1];
        if (valid_pose(pl, type, nx, ny, final)) {
            loc->x = nx; loc->y = ny; loc->rot = final;
            if (type == PT) {
                loc->rot_flag = 1;
                loc->last_kick_flag = (ki == kl->n - 1 && dir != 2);
            }
            return 1;
        }
    }
    return 0;
}

/* MoveGenerator._brute_force_algorithm -- move_generation.py:204-243 */
static void brute_force(movegen *mg, int start_x, int start_y, int check_rotations) {
    static uint8_t checked[TRL_PLANES][TRL_POLICY_ROWS][4];     /* checked_list[x+2][y][r], :155-162 */
    memset(checked, 0, sizeof checked);
    int qcap = 1 << 16, qh = 0, qt = 0;
    pose *q = (pose *)malloc(sizeof(pose) * (size_t)qcap);
    q[qt++] = (pose){start_x, start_y, 0, 0, 0};
    const to_player *pl = mg->board;
    int type = mg->type;
    while (qh < qt) {
        pose loc = q[qh++];
        int down_ok = valid_pose(pl, type, loc.x, loc.y + 1, loc.rot);
        if (checked[loc.x + 2][loc.y][loc.rot]) {                            /* :212-217 */
            if (!down_ok) {
                int present = 0;
                for (int i = 0; i < mg->n_place; i++) {
                    placement *p = &mg->place[i];
                    if (p->x == loc.x && p->y == loc.y && p->rot == loc.rot &&
                        p->flag == loc.rot_flag && p->last == loc.last_kick_flag) { present = 1; break; }
                }
                if (!present) add_place(mg, loc.x, loc.y, loc.rot, loc.rot_flag, loc.last_kick_flag);
            }
            continue;
        }
        checked[loc.x + 2][loc.y][loc.rot] = 1;
        if (!down_ok) add_place(mg, loc.x, loc.y, loc.rot, loc.rot_flag, loc.last_kick_flag);
        static const int mv[3][2] = {{1,0},{-1,0},{0,1}};                    /* :225-232 */
        for (int i = 0; i < 3; i++)
            if (valid_pose(pl, type, loc.x + mv[i][0], loc.y + mv[i][1], loc.rot)) {
                if (qt == qcap) { qcap *= 2; q = (pose *)realloc(q, sizeof(pose) * (size_t)qcap); }
                q[qt++] = (pose){loc.x + mv[i][0], loc.y + mv[i][1], loc.rot, 0, 0};
            }
        if (check_rotations)                                                 /* :235-243 */
            for (int d = 1; d < 4; d++) {
                pose nl = loc;
                if (try_wallkick(pl, type, &nl, d) && nl.y >= 0) {
                    if (qt == qcap) { qcap *= 2; q = (pose *)realloc(q, sizeof(pose) * (size_t)qcap); }
                    q[qt++] = nl;
                }
            }
    }
    free(q);
}

/* MoveGenerator._convert_placements_to_policy -- move_generation.py:650-749 */
static void convert_placements(const movegen *mg, uint8_t *mask) {
    /* unique_placements: key (x,y,rot) -> 0 absent, 1 unflagged, 2 flagged, 3 flagged+last kick;
     * "if key not in unique_placements or loc.flag: unique[key] = loc" */
    static uint8_t uniq[MAP_W][MAP_H][4];
    memset(uniq, 0, sizeof uniq);
    for (int i = 0; i < mg->n_place; i++) {
        const placement *p = &mg->place[i];
        uint8_t *u = &uniq[p->x + 2][p->y + 2][p->rot];
        if (*u == 0 || p->flag) *u = (uint8_t)(p->flag ? (p->last ? 3 : 2) : 1);
    }
    int type = mg->type, is_T = (type == PT), is_ZSI = (type == PZ || type == PS || type == PI);
    for (int xm = 0; xm < MAP_W; xm++)
        for (int ym = 0; ym < MAP_H; ym++)
            for (int o = 0; o < 4; o++) {
                uint8_t u = uniq[xm][ym][o];
                if (!u) continue;
                int ts = 0;
                if (is_T && u >= 2) ts = (u == 3) ? 2 : 1;
                int pi = policy_index(type, o % N_ROT[type], ts);
                int new_col = xm - 2, new_row = ym - 2;
                if (is_ZSI) { if (o == 2) new_row += 1; else if (o == 3) new_col -= 1; }
                /* the reference asserts new_row >= 0 (:706-710); y < 0 never reaches here */
                mask[(pi * TRL_POLICY_ROWS + new_row) * TRL_POLICY_COLS + (new_col + 2)] = 1;
            }
}

/* MoveGenerator._generate_moves_for_piece -- move_generation.py:107-153 */
static void moves_for_piece(const to_player *player, int piece_type, int algo, uint8_t *mask) {
    to_player sim = *player;            /* copy_no_board(): the board is never written below */
    if (sim.piece == TRL_NO_PIECE || sim.piece != piece_type) hold_piece(&sim);
    if (sim.piece == TRL_NO_PIECE) return;
    movegen mg;
    mg.board = player; mg.type = sim.piece; mg.n_place = 0;
    mg.place = (placement *)malloc(sizeof(placement) * MAX_PLACE);
    int check_rotations = (mg.type != PO);
    /* _set_starting_position -- :164-172 */
    int start_y = imax(highest_row(player) - BOX[mg.type], SPAWN_Y);
    int start_x = spawn_x(mg.type);
    if (algo == 0) convolutional(&mg, start_x, start_y, check_rotations);
    else brute_force(&mg, start_x, start_y, check_rotations);
    convert_placements(&mg, mask);
    free(mg.place);
}

int to_move_matrix(const to_player *pl, int ruleset, int algo, uint8_t *mask) {
    (void)ruleset;
    memset(mask, 0, TRL_POLICY_SIZE);
    /* _get_piece_types_to_check -- move_generation.py:91-105 */
    int p1 = pl->piece, p2;
    if (pl->held != TRL_NO_PIECE) p2 = pl->held;
    else if (pl->queue_len > 0) p2 = pl->queue[0];
    else p2 = TRL_NO_PIECE;
    if (p1 == p2) { if (p1 != TRL_NO_PIECE) moves_for_piece(pl, p1, algo, mask); }
    else {
        if (p1 != TRL_NO_PIECE) moves_for_piece(pl, p1, algo, mask);
        if (p2 != TRL_NO_PIECE) moves_for_piece(pl, p2, algo, mask);
    }
    int n = 0;
    for (int i = 0; i < TRL_POLICY_SIZE; i++) n += mask[i];
    return n;
}

int to_legal_moves(const to_game *g, uint16_t *out) { return to_legal_moves_algo(g, 0, out); }

/* get_move_matrix(player, algo=config.move_algorithm) + get_move_list (ai.py:414-417): algo 0 'convolutional', 1 'brute-force' */
int to_legal_moves_algo(const to_game *g, int algo, uint16_t *out) {
    if (to_is_terminal(g) || to_no_move(g)) return 0;
    uint8_t mask[TRL_POLICY_SIZE];
    to_move_matrix(&g->p[g->turn], g->ruleset, algo, mask);
    int n = 0;
    for (int pi = 0; pi < TRL_PLANES; pi++)                 /* np.argwhere order, ai.py:1019 */
        for (int r = 0; r < TRL_POLICY_ROWS; r++)
            for (int c = 0; c < TRL_POLICY_COLS; c++)
                if (mask[(pi * TRL_POLICY_ROWS + r) * TRL_POLICY_COLS + c]) {
                    if (n < TRL_MAX_LEGAL) out[n] = TRL_MOVE(pi, r, c);
                    n++;
                }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * canonical export / import
 * ---------------------------------------------------------------------------------------- */
void to_export(const to_game *g, trl_game_state *out) {
    memset(out, 0, sizeof *out);
    int ovf = g->overflow;
    for (int i = 0; i < 2; i++) {
        const to_player *p = &g->p[i];
        trl_player_state *o = &out->p[i];
        for (int r = 0; r < TRL_ROWS; r++) {
            uint16_t bits = 0;
            for (int c = 0; c < TRL_COLS; c++) if (p->grid[r][c]) bits |= (uint16_t)(1u << c);
            o->rows[r] = bits;
        }
        if (p->garbage_recv_len > TRL_MAX_GARBAGE || p->queue_len > TRL_MAX_QUEUE) ovf = 1;
        int gl = imin(p->garbage_recv_len, TRL_MAX_GARBAGE);
        for (int k = 0; k < gl; k++) o->garbage[k >> 4] |= (uint64_t)p->garbage_recv[k] << (4 * (k & 15));
        int ql = imin(p->queue_len, TRL_MAX_QUEUE);
        for (int k = 0; k < ql; k++) o->queue |= (uint64_t)p->queue[k] << (4 * k);
        o->pieces = p->pieces; o->lines_sent = p->lines_sent; o->lines_cleared = p->lines_cleared;
        o->b2b = (int16_t)p->b2b; o->combo = (int16_t)p->combo; o->b2b_level = (uint8_t)p->b2b_level;
        o->piece = (uint8_t)p->piece; o->held = (uint8_t)p->held;
        o->queue_len = (uint8_t)ql; o->garbage_len = (uint8_t)gl; o->game_over = (uint8_t)p->game_over;
    }
    out->stream = g->stream; out->bag_ctr = g->bag_ctr; out->garb_ctr = g->garb_ctr;
    out->ply = g->ply; out->pol_ctr = g->pol_ctr; out->episode = g->episode;
    out->turn = (uint8_t)g->turn; out->flags = (uint8_t)(ovf ? TRL_FLAG_OVERFLOW : 0);
    out->ruleset = (uint8_t)g->ruleset;
}

void to_import(to_game *g, const trl_game_state *in, uint64_t seed) {
    memset(g, 0, sizeof *g);
    for (int i = 0; i < 2; i++) {
        to_player *p = &g->p[i];
        const trl_player_state *o = &in->p[i];
        for (int r = 0; r < TRL_ROWS; r++)
            for (int c = 0; c < TRL_COLS; c++) p->grid[r][c] = (o->rows[r] >> c) & 1;
        p->garbage_recv_len = o->garbage_len;
        for (int k = 0; k < o->garbage_len; k++) p->garbage_recv[k] = (int)((o->garbage[k >> 4] >> (4 * (k & 15))) & 15);
        p->queue_len = o->queue_len;
        for (int k = 0; k < o->queue_len; k++) p->queue[k] = (int)((o->queue >> (4 * k)) & 15);
        p->pieces = o->pieces; p->lines_sent = o->lines_sent; p->lines_cleared = o->lines_cleared;
        p->b2b = o->b2b; p->combo = o->combo; p->b2b_level = o->b2b_level;
        p->piece = o->piece; p->held = o->held; p->game_over = o->game_over;
    }
    g->seed = seed;
    g->stream = in->stream; g->bag_ctr = in->bag_ctr; g->garb_ctr = in->garb_ctr;
    g->ply = in->ply; g->pol_ctr = in->pol_ctr; g->episode = in->episode;
    g->turn = in->turn; g->ruleset = in->ruleset; g->overflow = (in->flags & TRL_FLAG_OVERFLOW) != 0;
}

/* ------------------------------------------------------------------------------------------
 * C2 workload on the CPU (same definition as trl_env_rollout_random)
 * ---------------------------------------------------------------------------------------- */
uint64_t to_rollout_random(to_game *g, int n_plies, int max_plies) {
    uint64_t steps = 0;
    uint16_t moves[TRL_MAX_LEGAL];
    for (int k = 0; k < n_plies; k++) {
        int n = to_legal_moves(g, moves);
        if (n > TRL_MAX_LEGAL) { g->overflow = 1; n = TRL_MAX_LEGAL; }
        if (n > 0) {
            int idx = rng_below(g, TRL_RNG_POLICY, g->pol_ctr++, (uint32_t)n);
            uint16_t m = moves[idx];
            to_make_move(g, TRL_MOVE_PLANE(m), TRL_MOVE_COL(m) - 2, TRL_MOVE_ROW(m), 1);
            steps++;
        }
        if (n == 0 || to_is_terminal(g) || (int)g->ply >= max_plies) {
            /* auto-reset: Game.setup() on a fresh stream; the slot keeps its identity */
            uint64_t seed = g->seed; uint32_t stream = g->stream, episode = g->episode + 1;
            int ruleset = g->ruleset, ovf = g->overflow;
            to_game_setup(g, ruleset, seed, stream + 0x01000000u);
            g->episode = episode; g->overflow = ovf;
        }
    }
    return steps;
}

void to_batch_setup(to_game *gs, int n, int ruleset, uint64_t seed, const uint32_t *streams) {
    for (int i = 0; i < n; i++) to_game_setup(&gs[i], ruleset, seed, streams ? streams[i] : (uint32_t)i);
}
uint64_t to_batch_rollout_random(to_game *gs, int n, int n_plies, int max_plies) {
    uint64_t s = 0;
    for (int i = 0; i < n; i++) s += to_rollout_random(&gs[i], n_plies, max_plies);
    return s;
}
void to_batch_export(const to_game *gs, int n, trl_game_state *out) {
    for (int i = 0; i < n; i++) to_export(&gs[i], &out[i]);
}
size_t to_sizeof_game(void) { return sizeof(to_game); }

/* helpers for the ctypes test harness */
const to_player *to_player_ptr(const to_game *g, int i) { return &g->p[i]; }

/* rows int16[n][8]: ruleset, rows_cleared, tspin, mini, all_clear, combo, b2b, level
 * out  int16[n][4]: attack, combo', b2b', level'   (exhaustive check of stats.py:45-157) */
void to_attack_table(const int16_t *rows, int n, int16_t *out) {
    for (int i = 0; i < n; i++) {
        const int16_t *r = rows + 8 * i;
        to_player p; memset(&p, 0, sizeof p);
        p.combo = r[5]; p.b2b = r[6]; p.b2b_level = r[7];
        int a = to_get_attack(&p, r[0], r[1], r[2], r[3], r[4]);
        out[4 * i + 0] = (int16_t)a; out[4 * i + 1] = (int16_t)p.combo;
        out[4 * i + 2] = (int16_t)p.b2b; out[4 * i + 3] = (int16_t)p.b2b_level;
    }
}
