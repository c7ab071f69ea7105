/*
 * tetris_oracle_mcts.c -- CPU restatement of the reference search (ai.py:237-659, 1016-1024,
 * 1330-1361).  TEST INFRASTRUCTURE ONLY (see tetris_oracle.h).
 *
 * Arithmetic.  The reference mixes Python floats (fp64) and numpy float32 scalars; under
 * the numpy >= 2 promotion rules of the build container (numpy 2.2.6, NEP 50: Python scalars
 * are "weak") a non-root node's priors are float32 and every expression that touches one is
 * evaluated in float32, while root priors (math.exp of the root softmax, ai.py:428-434, and
 * the float64 gamma noise, ai.py:482-499), value sums and Q stay fp64.  This file follows
 * that op by op; the casts below are exactly the points where numpy rounds to float32.
 * (Under the reference's pinned numpy 1.24.2 scalar-scalar products promote to fp64 instead;
 * that variant cannot be run here and is not pinned.)
 */
#include "tetris_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- random draws of the search (engine definition, include/tetris_engine.h) ----------------
 * purpose 3: playout-cap coin, one per search, counter = ply of the root position
 * purpose 4: temperature pick, one per search, counter = ply
 * purpose 5: gamma noise, counter = ((ply * 512 + child) * 64 + attempt) * 4 + k
 * uniform double in (0,1): ((u64 >> 11) + 0.5) * 2^-53                                        */
static uint64_t mix64m(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_unif(uint64_t seed, uint32_t stream, uint32_t purpose, uint64_t ctr) {
    uint64_t k = mix64m(seed ^ ((uint64_t)stream * 0xD1342543DE82EF95ull));
    uint64_t u = mix64m(k ^ ((uint64_t)purpose << 56) ^ ctr);
    return ((double)(u >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
/* gamma(shape a, scale 1), Marsaglia-Tsang on a + 1 with the u^(1/a) boost (engine definition) */
double to_gamma(uint64_t seed, uint32_t stream, uint64_t base_ctr, double a) {
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
    return 0.0;
}

/* numpy's pairwise float64 sum (np.sum of a contiguous 1-D array) */
static double np_sum(const double *a, int n) {
    if (n < 8) { double r = 0.0; for (int i = 0; i < n; i++) r += a[i]; return r; }
    if (n <= 128) {
        double r[8]; int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    int n2 = n / 2; n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

/* ---- tree (MCTSNode / MCTSTree / NodeState, ai.py:237-297) ---------------------------------- */
typedef struct {
    int parent, first_child, n_children;
    int move;                 /* encoded, -1 for the root */
    int visit_count;
    double value_sum, value_avg;
    double policy;            /* holds the float32 value exactly when the prior is float32-typed */
    int prior_f32;            /* type of this node's prior: 1 = numpy float32, 0 = fp64 */
    int state;                /* index into states[], -1 while node.game is None */
} mnode;

typedef struct {
    mnode *nodes; int n_nodes, cap_nodes;
    to_game *states; int n_states, cap_states;
} mtree;

static int new_node(mtree *t, int parent, int move) {
    if (t->n_nodes == t->cap_nodes) {
        t->cap_nodes *= 2;
        t->nodes = (mnode *)realloc(t->nodes, sizeof(mnode) * (size_t)t->cap_nodes);
    }
    mnode *n = &t->nodes[t->n_nodes];
    memset(n, 0, sizeof *n);
    n->parent = parent; n->move = move; n->first_child = -1; n->state = -1;
    return t->n_nodes++;
}
static int new_state(mtree *t) {
    if (t->n_states == t->cap_states) {
        t->cap_states *= 2;
        t->states = (to_game *)realloc(t->states, sizeof(to_game) * (size_t)t->cap_states);
    }
    return t->n_states++;
}

/* Player.copy (player.py:217-233): lines_sent / lines_cleared / garbage_to_send are not copied */
static void game_copy(to_game *dst, const to_game *src) {
    *dst = *src;
    for (int i = 0; i < 2; i++) { dst->p[i].lines_sent = 0; dst->p[i].lines_cleared = 0; dst->p[i].garbage_send_len = 0; }
}

int to_mcts(to_game *game, const to_mcts_config *cfg, to_eval_fn eval, void *ctx, to_mcts_result *out) {
    mtree T;
    T.cap_nodes = 1 << 14; T.n_nodes = 0; T.nodes = (mnode *)malloc(sizeof(mnode) * (size_t)T.cap_nodes);
    T.cap_states = 512; T.n_states = 0; T.states = (to_game *)malloc(sizeof(to_game) * (size_t)T.cap_states);
    float *policy = (float *)malloc(sizeof(float) * TRL_POLICY_SIZE);
    /* Config.value_min / value_mid / value_max / negate_value (ai.py:198-216) */
    const double value_min = cfg->use_tanh ? -1.0 : 0.0, value_max = 1.0, value_mid = cfg->use_tanh ? 0.0 : 0.5;
#define NEGATE(v) (cfg->use_tanh ? -(v) : 1.0 - (v))

    /* root: game.copy() with previews restricted to 5 (ai.py:304-313) */
    int root = new_node(&T, -1, -1);
    T.nodes[root].state = new_state(&T);
    game_copy(&T.states[0], game);
    for (int i = 0; i < 2; i++) if (T.states[0].p[i].queue_len > TRL_PREVIEWS) T.states[0].p[i].queue_len = TRL_PREVIEWS;

    /* playout-cap randomisation (ai.py:322-330) */
    int max_iterations, fast_iter = 0;
    if (cfg->training && cfg->use_playout_cap) {
        double denom = cfg->playout_cap_chance * (cfg->playout_cap_mult - 1) + 1;
        if (rng_unif(game->seed, game->stream, 3, game->ply) < cfg->playout_cap_chance)
            max_iterations = (int)ceil(cfg->playout_cap_mult * (cfg->max_iter / denom));
        else { max_iterations = (int)floor(cfg->max_iter / denom); fast_iter = 1; }
    } else max_iterations = cfg->max_iter;

    uint16_t moves[TRL_MAX_LEGAL];
    for (int iter = 1; iter <= max_iterations; iter++) {
        int node = root;
        /* SELECT (ai.py:346-393) */
        while (T.nodes[node].n_children > 0) {
            mnode *nd = &T.nodes[node];
            int parent_visits = nd->visit_count;
            double sqrt_parent = sqrt((double)parent_visits);
            double unvisited_U_scale = cfg->cpuct * sqrt_parent / cfg->dpuct;
            int check_forced = cfg->use_forced && cfg->training && node == root && !(cfg->use_playout_cap && fast_iter);
            int best = -1; int have = 0; double best_d = -1.0; float best_f = 0; int best_is_f = 0;
            for (int k = 0; k < nd->n_children; k++) {
                mnode *ch = &T.nodes[nd->first_child + k];
                int vc = ch->visit_count;
                int is_f = ch->prior_f32;
                double score_d = 0; float score_f = 0;
                if (is_f) {                       /* float32 expression (weak Python scalars) */
                    float p = (float)ch->policy, U;
                    if (vc == 0) U = (float)unvisited_U_scale * p;
                    else U = (((float)cfg->cpuct * p) * (float)sqrt_parent) / (float)(cfg->dpuct + vc);
                    score_f = (float)ch->value_avg + U;
                } else {
                    double U;
                    if (vc == 0) U = unvisited_U_scale * ch->policy;
                    else U = cfg->cpuct * ch->policy * sqrt_parent / (cfg->dpuct + vc);
                    score_d = ch->value_avg + U;
                }
                if (check_forced && vc >= 1) {    /* ai.py:377-381 */
                    double n_forced = sqrt(cfg->c_forced * ch->policy * parent_visits);
                    if (vc < n_forced) { score_d = INFINITY; score_f = INFINITY; }
                }
                /* child_score >= max_child_score, max starts at -1 (ai.py:348,383-385) */
                double s = is_f ? (double)score_f : score_d;
                double m = !have ? -1.0 : (best_is_f ? (double)best_f : best_d);
                if (s >= m) { have = 1; best = nd->first_child + k; best_is_f = is_f; best_f = score_f; best_d = score_d; }
            }
            node = best;
        }
        int playout = node;
        /* STEP (ai.py:398-403) */
        if (node != root) {
            mnode *nd = &T.nodes[node];
            if (nd->state < 0) { int si = new_state(&T); T.nodes[node].state = si; nd = &T.nodes[node]; }
            to_game *g = &T.states[nd->state];
            game_copy(g, &T.states[T.nodes[nd->parent].state]);
            g->garb_ctr = game->garb_ctr;          /* the reference draws from one global stream */
            to_make_move(g, TRL_MOVE_PLANE(nd->move), TRL_MOVE_COL(nd->move) - 2, TRL_MOVE_ROW(nd->move), 0);
            game->garb_ctr = g->garb_ctr;
        }
        to_game *lg = &T.states[T.nodes[node].state];
        double value;
        if (!to_is_terminal(lg)) {
            /* EVAL + EXPAND (ai.py:406-458) */
            float v32;
            eval(lg, &v32, policy, ctx);
            value = (double)v32;
            if (!to_no_move(lg)) {
                int n = to_legal_moves_algo(lg, cfg->move_algorithm, moves);
                int is_root = (node == root);
                int first = T.n_nodes;
                if (is_root && cfg->use_root_softmax) {
                    double *pol = (double *)malloc(sizeof(double) * (size_t)n);
                    float maxp = 0;
                    for (int k = 0; k < n; k++) {
                        int pl = TRL_MOVE_PLANE(moves[k]), r = TRL_MOVE_ROW(moves[k]), c = TRL_MOVE_COL(moves[k]);
                        float p = policy[(pl * TRL_POLICY_ROWS + r) * TRL_POLICY_COLS + c];
                        if (p <= 0) p = 1e-25f;                             /* ai.py:411 */
                        pol[k] = (double)p;
                        if (k == 0 || p > maxp) maxp = p;
                    }
                    double log_max = log((double)maxp), inv_temp = 1.0 / cfg->root_softmax_temp;
                    double sum = 0.0;
                    for (int k = 0; k < n; k++) { pol[k] = exp((log(pol[k]) - log_max) * inv_temp); sum += pol[k]; }
                    for (int k = 0; k < n; k++) {
                        int c = new_node(&T, node, moves[k]);
                        T.nodes[c].policy = pol[k] / sum; T.nodes[c].prior_f32 = 0;
                    }
                    free(pol);
                } else {
                    float sum = 0.0f;
                    float *pol = (float *)malloc(sizeof(float) * (size_t)n);
                    for (int k = 0; k < n; k++) {
                        int pl = TRL_MOVE_PLANE(moves[k]), r = TRL_MOVE_ROW(moves[k]), c = TRL_MOVE_COL(moves[k]);
                        float p = policy[(pl * TRL_POLICY_ROWS + r) * TRL_POLICY_COLS + c];
                        if (p <= 0) p = 1e-25f;
                        pol[k] = p; sum = sum + p;                          /* sum() of np.float32 scalars */
                    }
                    for (int k = 0; k < n; k++) {
                        int c = new_node(&T, node, moves[k]);
                        T.nodes[c].policy = (double)(pol[k] / sum); T.nodes[c].prior_f32 = 1;
                    }
                    free(pol);
                }
                T.nodes[node].first_child = first; T.nodes[node].n_children = n;
                for (int k = 0; k < n; k++) {                               /* FPU init (ai.py:447-453) */
                    mnode *c = &T.nodes[first + k];
                    if (cfg->fpu_reduction) { double pv = NEGATE(value); c->value_avg = pv > value_min ? pv : value_min; }
                    else c->value_avg = cfg->fpu_value > value_min ? cfg->fpu_value : value_min;
                }
            }
        } else {                                                            /* ai.py:472-479 */
            int w = to_winner(lg);
            if (w == lg->turn) value = value_max;
            else if (w == 1 - lg->turn) value = value_min;
            else value = value_mid;
        }
        /* NOISE (ai.py:482-499): un-normalised gamma draws, only when the playout node is the root */
        if (cfg->training && !fast_iter && cfg->use_dirichlet && node == root) {
            int n = T.nodes[root].n_children;
            double a = cfg->dir_alpha;
            if (cfg->use_dir_s) a *= cfg->dir_s / n;
            for (int k = 0; k < n; k++) {
                mnode *c = &T.nodes[T.nodes[root].first_child + k];
                double noise = to_gamma(game->seed, game->stream, (uint64_t)game->ply * 512 + (uint64_t)k, a);
                if (c->prior_f32)     /* np.float32 * Python float stays float32, the sum with the float64 noise is fp64 */
                    c->policy = (double)((float)c->policy * (float)(1 - cfg->dir_expl)) + noise * cfg->dir_expl;
                else
                    c->policy = c->policy * (1 - cfg->dir_expl) + noise * cfg->dir_expl;
                c->prior_f32 = 0;
            }
        }
        /* BACKUP (ai.py:504-533) */
        value = NEGATE(value);
        double pos_value = value, neg_value = NEGATE(value);
        int final_turn = lg->turn;
        for (int n = node; ; n = T.nodes[n].parent) {
            mnode *nd = &T.nodes[n];
            nd->visit_count += 1;
            nd->value_sum += (T.states[nd->state].turn == final_turn) ? pos_value : neg_value;
            nd->value_avg = nd->value_sum / nd->visit_count;
            if (n == root) break;
        }
        /* FPU refresh of the unvisited siblings (ai.py:536-565) */
        if (playout != root && cfg->fpu_reduction) {
            int parent = T.nodes[playout].parent;
            if (parent != root) {
                mnode *pn = &T.nodes[parent];
                float explored = 0.0f; int any = 0;
                for (int k = 0; k < pn->n_children; k++) {
                    mnode *c = &T.nodes[pn->first_child + k];
                    if (c->visit_count > 0) { explored = any ? explored + (float)c->policy : (float)c->policy; any = 1; }
                }
                double fpu = NEGATE(pn->value_avg) - cfg->fpu_value * sqrt((double)explored);
                if (!(fpu > value_min)) fpu = value_min;
                for (int k = 0; k < pn->n_children; k++) {
                    mnode *c = &T.nodes[pn->first_child + k];
                    if (c->visit_count == 0) c->value_avg = fpu;
                }
            }
        }
    }

    /* choose the move (ai.py:568-614) */
    mnode *rt = &T.nodes[root];
    int n = rt->n_children;
    int max_n = 0, max_k = -1;
    for (int k = 0; k < n; k++) { int v = T.nodes[rt->first_child + k].visit_count; if (v >= max_n) { max_n = v; max_k = k; } }
    int sel = 0;
    double temp = cfg->training ? cfg->temperature : 0.0;
    if (n > 0) {
        if (temp == 0.0) {
            int bestv = -1;
            for (int k = 0; k < n; k++) { int v = T.nodes[rt->first_child + k].visit_count; if (v > bestv) { bestv = v; sel = k; } }   /* np.argmax */
        } else {
            double *probs = (double *)malloc(sizeof(double) * (size_t)n);
            for (int k = 0; k < n; k++) probs[k] = pow((double)T.nodes[rt->first_child + k].visit_count, 1.0 / temp);
            double s = np_sum(probs, n);
            for (int k = 0; k < n; k++) probs[k] = probs[k] / s;
            /* random.choices(range(n), weights=probs): accumulate, bisect_right over [0, n-1) */
            double acc = 0.0;
            for (int k = 0; k < n; k++) { acc = (k == 0) ? probs[0] : acc + probs[k]; probs[k] = acc; }
            double total = probs[n - 1] + 0.0;
            double x = rng_unif(game->seed, game->stream, 4, game->ply) * total;
            int lo = 0, hi = n - 1;
            while (lo < hi) { int mid = (lo + hi) / 2; if (x < probs[mid]) hi = mid; else lo = mid + 1; }
            sel = lo;
            free(probs);
        }
    }
    /* policy-target pruning AFTER the move is chosen (ai.py:616-648): visit counts only */
    if (cfg->use_forced && cfg->training && !fast_iter && n > 0) {
        mnode *mp = &T.nodes[rt->first_child + max_k];          /* max_id: the LAST child with the largest N */
        double most = mp->value_avg + cfg->cpuct * mp->policy * sqrt((double)rt->visit_count) / (cfg->dpuct + mp->visit_count);
        for (int k = 0; k < n; k++) {
            if (k == max_k) continue;
            mnode *c = &T.nodes[rt->first_child + k];
            if (c->visit_count > 0) {
                double n_forced = sqrt(cfg->c_forced * c->policy * rt->visit_count);
                int count = 0;
                for (;;) {
                    if (c->visit_count == 1) { c->visit_count = 0; break; }
                    /* the reference puts the ROOT's N in the denominator here (ai.py:640) */
                    double minus = c->value_avg + cfg->cpuct * c->policy * sqrt((double)rt->visit_count) / (cfg->dpuct + rt->visit_count);
                    if (count < n_forced && minus < most) { count += 1; c->visit_count -= 1; }
                    else break;
                }
            }
        }
    }
    out->move = n > 0 ? T.nodes[rt->first_child + sel].move : -1;
    out->save_move = !fast_iter;
    out->max_iterations = max_iterations;
    out->n_root_children = n;
    out->n_nodes = T.n_nodes;
    out->root_visit = rt->visit_count;
    out->root_p_f32 = n > 0 ? T.nodes[rt->first_child].prior_f32 : 0;
#undef NEGATE
    for (int k = 0; k < n && k < TRL_MAX_LEGAL; k++) {
        mnode *c = &T.nodes[rt->first_child + k];
        out->root_moves[k] = (uint16_t)c->move; out->root_visits[k] = c->visit_count;
        out->root_q[k] = c->value_avg; out->root_p[k] = c->policy;
    }
    free(policy); free(T.nodes); free(T.states);
    return 0;
}

/* ---- deterministic stub evaluator shared by the fixtures, the oracle and the GPU tests --------
 * h = fold of the 36 board/meta words of both players + turn; value = top 24 bits of mix(h) / 2^24;
 * policy[i] = u*u with u = top 24 bits of mix(h ^ (i+1)*K) / 2^24, 0 when the low nibble is 0.    */
uint64_t to_stub_hash(const trl_game_state *s) {
    uint64_t h = 0x243F6A8885A308D3ull;
    for (int i = 0; i < 2; i++) {
        const uint64_t *w = (const uint64_t *)&s->p[i];
        /* rows (10 words), garbage (4), queue (1); stats/pieces word by word below */
        for (int k = 0; k < 15; k++) h = mix64m(h ^ w[k]);
        const trl_player_state *p = &s->p[i];
        uint64_t meta = (uint64_t)(uint16_t)p->b2b | ((uint64_t)(uint16_t)p->combo << 16) | ((uint64_t)p->b2b_level << 32)
                      | ((uint64_t)p->piece << 40) | ((uint64_t)p->held << 44) | ((uint64_t)p->queue_len << 48)
                      | ((uint64_t)p->garbage_len << 53) | ((uint64_t)p->game_over << 60);
        h = mix64m(h ^ meta);
    }
    return mix64m(h ^ s->turn);
}
void to_stub_eval_state(const trl_game_state *s, float *value, float *policy) {
    uint64_t h = to_stub_hash(s);
    *value = (float)(mix64m(h) >> 40) * (1.0f / 16777216.0f);
    for (int i = 0; i < TRL_POLICY_SIZE; i++) {
        uint64_t k = mix64m(h ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull));
        float u = (float)(k >> 40) * (1.0f / 16777216.0f);
        policy[i] = (k & 15) == 0 ? 0.0f : u * u;
    }
}
void to_stub_eval(const to_game *g, float *value, float *policy, void *ctx) {
    (void)ctx;
    trl_game_state s;
    to_export(g, &s);
    to_stub_eval_state(&s, value, policy);
}

/* the stub for a tanh-range net (Config.use_tanh): value 2 v - 1 in [-1, 1), exact in float32 */
void to_stub_eval_tanh(const to_game *g, float *value, float *policy, void *ctx) {
    to_stub_eval(g, value, policy, ctx);
    *value = 2.0f * *value - 1.0f;
}

/* a second deterministic evaluator for the gating tests: the stub with a salted hash (ctx = &salt) */
void to_stub_eval_salted(const to_game *g, float *value, float *policy, void *ctx) {
    trl_game_state s;
    to_export(g, &s);
    uint64_t salt = ctx ? *(const uint64_t *)ctx : 0, h = to_stub_hash(&s);
    if (salt) h = mix64m(h ^ salt);
    *value = (float)(mix64m(h) >> 40) * (1.0f / 16777216.0f);
    for (int i = 0; i < TRL_POLICY_SIZE; i++) {
        uint64_t k = mix64m(h ^ ((uint64_t)(i + 1) * 0x9E3779B97F4A7C15ull));
        float u = (float)(k >> 40) * (1.0f / 16777216.0f);
        policy[i] = (k & 15) == 0 ? 0.0f : u * u;
    }
}

/* one game of battle_networks / aplay_battle_game (ai.py:2072-2085): evaluator `salt1` plays player `side`,
 * evaluator `salt2` the other; the searcher's evaluator is used for the whole search.  Returns plies. */
int to_play_battle_stub(to_game *g, const to_mcts_config *cfg, int side, uint64_t salt1, uint64_t salt2, int max_log,
                        uint16_t *ply_move, int *winner_out) {
    int plies = 0, history = 1;
    to_mcts_result *res = (to_mcts_result *)malloc(sizeof(to_mcts_result));
    while (!to_is_terminal(g) && history < TRL_MAX_MOVES) {
        uint64_t salt = ((g->turn == 0) == (side == 0)) ? salt1 : salt2;
        to_mcts(g, cfg, to_stub_eval_salted, &salt, res);
        if (res->move < 0) break;
        if (plies < max_log) ply_move[plies] = (uint16_t)res->move;
        int mover = g->turn;
        to_make_move(g, TRL_MOVE_PLANE(res->move), TRL_MOVE_COL(res->move) - 2, TRL_MOVE_ROW(res->move), 1);
        if (mover == 1 && g->turn == 0) history++;
        plies++;
    }
    *winner_out = to_winner(g);
    free(res);
    return plies;
}

int to_mcts_stub(to_game *game, const to_mcts_config *cfg, to_mcts_result *out) {
    return to_mcts(game, cfg, cfg->use_tanh ? to_stub_eval_tanh : to_stub_eval, NULL, out);
}

/* ---- play_game (ai.py:1570-1699) with the stub evaluator -----------------------------------------
 * One self-play game; per ply the root state before the search, the chosen move, the save flag
 * and the root children (move, visits).  history.states grows by one per full turn after the
 * initial entry (game.py:85-87), so the loop bound len(history.states) < MAX_MOVES is
 * 1 + (plies by player 1) < 1000.  Returns the number of plies, winner in *winner_out.          */
int to_play_game(to_game *g, const to_mcts_config *cfg, to_eval_fn eval, void *ctx, int max_log,
                 trl_game_state *ply_state, uint16_t *ply_move, int *ply_save, int *ply_nchild,
                 uint16_t *child_moves, int *child_visits, int max_children_total, int *winner_out) {
    int plies = 0, ctot = 0, history = 1;
    to_mcts_result *res = (to_mcts_result *)malloc(sizeof(to_mcts_result));
    if (cfg->random_start) {
        /* ai.py:1588-1608: num_random_moves = np.random.exponential(0.04 * DIRICHLET_S) (engine draw: purpose 6,
         * counter 0, -scale * log(u)); while it is > 0: a MAX_ITER = 1 search without playout cap, noise or forced
         * playouts, then pick_random_move_by_policy (ai.py:999-1014): random.choices(moves, policies) over the root
         * children's priors (engine draw: purpose 4, counter = ply).  Logged with save = 0 and nchild = 0. */
        double left = -(0.04 * cfg->dir_s) * log(rng_unif(g->seed, g->stream, 6, 0));
        to_mcts_config fast = *cfg;
        fast.max_iter = 1; fast.use_playout_cap = 0; fast.use_dirichlet = 0; fast.use_forced = 0;
        while (left > 0 && !to_is_terminal(g)) {
            if (plies < max_log) to_export(g, &ply_state[plies]);
            to_mcts(g, &fast, eval, ctx, res);
            int n = res->n_root_children;
            if (n <= 0) break;
            int sel;
            {   /* random.choices: itertools.accumulate, total = cum[-1] + 0.0, bisect_right(cum, x, 0, n - 1) */
                double *cum = (double *)malloc(sizeof(double) * (size_t)n);
                if (res->root_p_f32) { float a = 0; for (int k = 0; k < n; k++) { a = (k == 0) ? (float)res->root_p[0] : a + (float)res->root_p[k]; cum[k] = (double)a; } }
                else { double a = 0; for (int k = 0; k < n; k++) { a = (k == 0) ? res->root_p[0] : a + res->root_p[k]; cum[k] = a; } }
                double total = cum[n - 1] + 0.0;
                double x = rng_unif(g->seed, g->stream, 4, g->ply) * total;
                int lo = 0, hi = n - 1;
                while (lo < hi) { int mid = (lo + hi) / 2; if (x < cum[mid]) hi = mid; else lo = mid + 1; }
                sel = lo;
                free(cum);
            }
            int mv = res->root_moves[sel];
            if (plies < max_log) { ply_move[plies] = (uint16_t)mv; ply_save[plies] = 0; ply_nchild[plies] = 0; }
            int mover = g->turn;
            to_make_move(g, TRL_MOVE_PLANE(mv), TRL_MOVE_COL(mv) - 2, TRL_MOVE_ROW(mv), 1);
            if (mover == 1 && g->turn == 0) history++;
            plies++;
            left -= 1;
        }
    }
    while (!to_is_terminal(g) && history < TRL_MAX_MOVES) {
        if (plies < max_log) to_export(g, &ply_state[plies]);
        to_mcts(g, cfg, eval, ctx, res);
        if (res->move < 0) break;
        if (plies < max_log) {
            ply_move[plies] = (uint16_t)res->move; ply_save[plies] = res->save_move; ply_nchild[plies] = res->n_root_children;
            for (int k = 0; k < res->n_root_children && ctot < max_children_total; k++, ctot++) {
                child_moves[ctot] = res->root_moves[k]; child_visits[ctot] = res->root_visits[k];
            }
        }
        int mover = g->turn;
        to_make_move(g, TRL_MOVE_PLANE(res->move), TRL_MOVE_COL(res->move) - 2, TRL_MOVE_ROW(res->move), 1);
        if (mover == 1 && g->turn == 0) history++;
        plies++;
    }
    *winner_out = to_winner(g);
    free(res);
    return plies;
}
int to_play_game_stub(to_game *g, const to_mcts_config *cfg, int max_log, trl_game_state *ply_state, uint16_t *ply_move,
                      int *ply_save, int *ply_nchild, uint16_t *child_moves, int *child_visits, int max_children_total,
                      int *winner_out) {
    return to_play_game(g, cfg, cfg->use_tanh ? to_stub_eval_tanh : to_stub_eval, NULL, max_log, ply_state, ply_move, ply_save,
                        ply_nchild, child_moves, child_visits, max_children_total, winner_out);
}
