#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the Python reference.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs /root/reference); the GPU box
only ever sees the .npz files this writes.  Nothing from the reference's source text is
stored: fixtures are inputs (seeds, states, moves) and the outputs the reference computed.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz (deterministic)

Fixtures:
  env_traces.npz   per-ply canonical states + legal move lists of reference games driven by
                   (a) the uniform policy of the engine RNG and (b) a greedy 1-ply driver
  step_cases.npz   (state, move) -> state for every legal move of crafted / sampled states
                   (T-spins with and without the last-kick flag, all-clears, surge, garbage
                   cancel / spawn, spawn-blocked loss, hold paths)
  movegen.npz      legal masks (convolutional and brute-force) for the reference's own
                   test boards (util.py:34-37, top-padded to 40 rows) x 7 pieces, the
                   empty board, and sampled positions
  attack.npz       exhaustive Stats.get_attack table, both rulesets
"""
import ast
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 20260630
MINOS = rh.MINOS

mods = rh.import_reference()
Game = mods["game"].Game
Piece = mods["piece"].Piece
get_move_matrix = mods["move_generation"].get_move_matrix
Stats = mods["stats"].Stats
const = mods["const"]


# ---------------------------------------------------------------------------------------------
def new_game(ruleset, stream):
    rng = rh.EngineRNG(SEED, stream)
    rh.install_rng(mods, rng)
    g = Game(ruleset)
    g.setup()
    return g, rng


def legal_moves(g):
    """encoded legal moves in argwhere order (ai.py:1016-1024); [] when terminal/no move"""
    if g.is_terminal or g.no_move:
        return []
    mask = get_move_matrix(g.players[g.turn], "convolutional")
    return rh.mask_to_moves(mask)


def decode(m):
    return (m >> 10) & 31, (m & 15) - 2, (m >> 4) & 63      # plane, board col, row


def game_from_state(buf, seed=SEED):
    """Build reference Game objects from a canonical trl_game_state (bytes)."""
    import struct
    g = Game("s1" if buf[288 + 24 + 2] == 1 else "s2")
    tail = struct.unpack(rh.GAME_TAIL_FMT, buf[288:320])
    stream, bag_ctr, garb_ctr, ply, pol_ctr, episode, turn, flags, rs = tail
    rng = rh.EngineRNG(seed, stream)
    rng.bag_ctr, rng.garb_ctr, rng.pol_ctr = bag_ctr, garb_ctr, pol_ctr
    rh.install_rng(mods, rng)
    for i in range(2):
        f = struct.unpack(rh.PLAYER_FMT, buf[144 * i:144 * (i + 1)])
        rows, garb, q = f[0:40], f[40:44], f[44]
        pieces, lsent, lcleared, b2b, combo, lvl, piece, held, ql, gl, over = f[45:56]
        pl = g.players[i]
        for r in range(40):
            for c in range(10):
                pl.board.grid[r][c] = 1 if (rows[r] >> c) & 1 else 0
        pl.garbage_to_receive = [(garb[k >> 4] >> (4 * (k & 15))) & 15 for k in range(gl)]
        pl.queue.pieces = [MINOS[(q >> (4 * k)) & 15] for k in range(ql)]
        pl.stats.pieces, pl.stats.lines_sent, pl.stats.lines_cleared = pieces, lsent, lcleared
        pl.stats.b2b, pl.stats.combo, pl.stats.b2b_level = b2b, combo, lvl
        if piece != rh.NO_PIECE:
            pl.piece = Piece(type=MINOS[piece])
            pl.piece.move_to_spawn()
        else:
            pl.piece = None
        pl.held_piece = MINOS[held] if held != rh.NO_PIECE else None
        pl.game_over = bool(over)
    g.turn = turn
    return g, rng, ply


# ---------------------------------------------------------------------------------------------
# drivers
def holes_and_height(grid):
    holes, height = 0, 0
    for c in range(10):
        seen = False
        for r in range(40):
            if grid[r][c] != 0:
                if not seen:
                    height = max(height, 40 - r)
                seen = True
            elif seen:
                holes += 1
    return holes, height


def greedy_pick(g, rng, moves, flavour):
    """1-ply heuristic: maximise attack / clears, avoid holes; prefers spin planes."""
    best, best_s = None, None
    me = g.turn
    for k, m in enumerate(moves):
        snap = rng.snapshot()
        c = g.copy()
        c.players[me].stats.lines_sent = 0
        c.players[me].stats.lines_cleared = 0
        plane, col, row = decode(m)
        c.make_move((plane, col, row), add_bag=False, add_history=False)
        rng.restore(snap)
        p = c.players[me]
        holes, height = holes_and_height(p.board.grid)
        s = 8.0 * p.stats.lines_sent + 3.0 * p.stats.lines_cleared - 1.5 * holes - 0.3 * height
        if plane >= 19:
            s += 2.0 * flavour
        if p.game_over:
            s -= 100
        s += 1e-3 * ((k * 2654435761 + flavour * 97) % 1000) / 1000.0   # deterministic tie-break
        if best_s is None or s > best_s:
            best, best_s = m, s
    return best


def play(ruleset, stream, driver, max_plies, flavour=0):
    g, rng = new_game(ruleset, stream)
    states = [rh.pack_game(g, rng, 0, ruleset)]
    moves_played, legal_lists = [], []
    ply = 0
    while ply < max_plies:
        moves = legal_moves(g)
        legal_lists.append(moves)
        if not moves:
            break
        if driver == "uniform":
            m = moves[rng.policy_index(len(moves))]
        else:
            m = greedy_pick(g, rng, moves, flavour)
        plane, col, row = decode(m)
        g.make_move((plane, col, row))
        ply += 1
        moves_played.append(m)
        states.append(rh.pack_game(g, rng, ply, ruleset))
        if g.is_terminal:
            legal_lists.append([])
            break
    else:
        legal_lists.append(legal_moves(g))
    return states, moves_played, legal_lists


def ragged(lists, dtype=np.uint16):
    counts = np.array([len(x) for x in lists], dtype=np.int32)
    flat = np.array([v for x in lists for v in x], dtype=dtype)
    return counts, flat


# ---------------------------------------------------------------------------------------------
def gen_env_traces():
    games = []
    for ruleset in ("s1", "s2"):
        for stream in range(24):
            games.append((ruleset, stream, "uniform", 0))
        for stream in range(100, 116):
            games.append((ruleset, stream, "greedy", stream % 3))
    all_states, all_moves, all_legal = [], [], []
    meta = []   # ruleset, stream, driver, n_plies, state offset, legal offset
    cov = dict(sent=0, garbage_max=0, terminal=0)
    for ruleset, stream, driver, flavour in games:
        states, moves, legal = play(ruleset, stream, driver, 400 if driver == "greedy" else 150, flavour)
        meta.append((1 if ruleset == "s1" else 2, stream, 0 if driver == "uniform" else 1, len(moves),
                     len(all_states), len(all_legal)))
        all_states += states
        all_moves += moves
        all_legal += legal
        assert len(legal) == len(states)
        last = np.frombuffer(states[-1], dtype=np.uint8)
        cov["terminal"] += int(last[141] or last[144 + 141])
        import struct
        for s in states:
            for i in range(2):
                gl = s[144 * i + 140]
                cov["garbage_max"] = max(cov["garbage_max"], gl)
        f0 = struct.unpack(rh.PLAYER_FMT, states[-1][0:144])
        f1 = struct.unpack(rh.PLAYER_FMT, states[-1][144:288])
        cov["sent"] += f0[46] + f1[46]
    counts, flat = ragged(all_legal)
    np.savez_compressed(
        os.path.join(OUT, "env_traces.npz"),
        seed=np.uint64(SEED),
        meta=np.array(meta, dtype=np.int64),
        states=np.frombuffer(b"".join(all_states), dtype=np.uint8).reshape(-1, 320),
        moves=np.array(all_moves, dtype=np.uint16),
        legal_counts=counts, legal_flat=flat)
    print("env_traces: %d games, %d states, coverage %s" % (len(games), len(all_states), cov))
    return all_states


# ---------------------------------------------------------------------------------------------
def load_reference_boards():
    """The four literal boards of util.py:34-37 (26 rows), top-padded to 40 rows as the
    reference's own migration does (util.py:3129-3148).  Data only: parsed, not imported."""
    boards = {}
    with open(os.path.join(rh.REFERENCE_DIR, "util.py")) as f:
        for line in f:
            for name in ("util_t_spin_board", "util_z_spin_board", "util_move_algo_board", "util_move_algo_board_2"):
                if line.startswith(name + " ="):
                    grid = ast.literal_eval(line.split("=", 1)[1].strip())
                    pad = [[0] * 10 for _ in range(40 - len(grid))]
                    boards[name] = pad + grid
    assert len(boards) == 4
    return boards


def rows_of(grid):
    return [sum(1 << c for c in range(10) if grid[r][c] != 0) for r in range(40)]


def gen_movegen(sample_states):
    boards = load_reference_boards()
    boards["empty"] = [[0] * 10 for _ in range(40)]
    names = ["util_t_spin_board", "util_z_spin_board", "util_move_algo_board", "util_move_algo_board_2", "empty"]
    cases_rows, cases_piece, cases_held, cases_q0, conv, brute, has_brute = [], [], [], [], [], [], []

    def add(grid_rows, piece, held, q0, player, want_brute):
        cases_rows.append(grid_rows)
        cases_piece.append(piece); cases_held.append(held); cases_q0.append(q0)
        conv.append(rh.mask_to_moves(get_move_matrix(player, "convolutional")))
        if want_brute:
            brute.append(rh.mask_to_moves(get_move_matrix(player, "brute-force")))
        else:
            brute.append([])
        has_brute.append(1 if want_brute else 0)

    # (1) the reference's own test construction (tests.py:6-12, 38-43): board x piece, held = None,
    #     next piece = whatever the bag gave -> we pin it explicitly to each of the 7 letters' successor
    for name in names:
        for pt in MINOS:
            for held in (None, pt):          # held == piece -> single piece type (tests.py:23-32)
                g, rng = new_game("s1", 7)
                pl = g.players[g.turn]
                pl.board.grid = np.array(boards[name], dtype=object)
                pl.piece = Piece(type=pt)
                pl.piece.move_to_spawn()
                pl.held_piece = held
                q0 = MINOS[(MINOS.index(pt) + 3) % 7]
                pl.queue.pieces = [q0] + pl.queue.pieces[1:]
                add(rows_of(boards[name]), MINOS.index(pt), MINOS.index(held) if held else rh.NO_PIECE,
                    MINOS.index(q0), pl, True)
    n_ref = len(conv)
    # (2) sampled positions from the greedy games, both algorithms on a subset
    rs = np.random.RandomState(7)
    idx = rs.choice(len(sample_states), size=min(600, len(sample_states)), replace=False)
    for n, i in enumerate(sorted(idx)):
        g, rng, ply = game_from_state(sample_states[i])
        if g.is_terminal or g.no_move:
            continue
        pl = g.players[g.turn]
        add(rows_of(pl.board.grid), MINOS.index(pl.piece.type) if pl.piece else rh.NO_PIECE,
            MINOS.index(pl.held_piece) if pl.held_piece else rh.NO_PIECE,
            MINOS.index(pl.queue.pieces[0]) if pl.queue.pieces else rh.NO_PIECE, pl, n % 4 == 0)
    cc, cf = ragged(conv)
    bc, bf = ragged(brute)
    np.savez_compressed(
        os.path.join(OUT, "movegen.npz"),
        n_reference_cases=np.int32(n_ref),
        rows=np.array(cases_rows, dtype=np.uint16), piece=np.array(cases_piece, dtype=np.uint8),
        held=np.array(cases_held, dtype=np.uint8), q0=np.array(cases_q0, dtype=np.uint8),
        conv_counts=cc, conv_flat=cf, brute_counts=bc, brute_flat=bf, has_brute=np.array(has_brute, dtype=np.uint8))
    # known answers of the reference's tests
    diff = sum(1 for a, b, h in zip(conv, brute, has_brute) if h and a != b)
    print("movegen: %d cases (%d reference-board cases), conv!=brute on %d" % (len(conv), n_ref, diff))


# ---------------------------------------------------------------------------------------------
def crafted_states():
    """Positions that force the rare branches of player.py:109-188 / game.py:100-117."""
    out = []

    def base(ruleset, stream):
        g, rng = new_game(ruleset, stream)
        return g, rng

    def fill(pl, rows_spec):
        # rows_spec: {row: string of 10 chars, 'X' filled}
        for r, s in rows_spec.items():
            for c, ch in enumerate(s):
                pl.board.grid[r][c] = 1 if ch == "X" else 0

    for ruleset in ("s1", "s2"):
        # all-clear with I flat / vertical, O, and a perfect-clear double
        for k, (spec, piece) in enumerate([
            ({39: "XXXXXX...." }, "I"), ({39: "....XXXXXX"}, "I"),
            ({39: "XXXXXXXX..", 38: "XXXXXXXX.."}, "O"),
            ({39: "XXXXXXXXX.", 38: "XXXXXXXXX.", 37: "XXXXXXXXX.", 36: "XXXXXXXXX."}, "I"),
            ({39: "XXX...XXXX", 38: "XXXX.XXXXX"}, "T"),
        ]):
            g, rng = base(ruleset, 200 + k)
            pl = g.players[0]
            fill(pl, spec)
            pl.piece = Piece(type=piece); pl.piece.move_to_spawn()
            pl.stats.b2b = 5; pl.stats.b2b_level = 2; pl.stats.combo = 3     # surge / b2b branches
            out.append((g, rng, ruleset))
        # T-spin double / triple / mini slots, both orientations, with pending garbage both ways
        tslots = [
            {39: "XX.XXXXXXX", 38: "X...XXXXXX", 37: "XX..XXXXXX"},            # TSD left overhang
            {39: "XXXXXXX.XX", 38: "XXXXXX...X", 37: "XXXXXX..XX"},            # TSD right overhang
            {39: "XXXXXXXX.X", 38: "XXXXXXX..X", 37: "XXXXXXXX.X", 36: "XXXXXXX..X", 35: "XXXXXXX.XX"},  # TST-like
            {39: "X.XXXXXXXX", 38: "X..XXXXXXX", 37: "X.XXXXXXXX", 36: "X..XXXXXXX", 35: "XX.XXXXXXX"},
            {39: "XXX.XXXXXX", 38: "XX...XXXXX"},                              # open T hole (mini / no spin)
            {39: ".XXXXXXXXX", 38: "..XXXXXXXX", 37: "X.XXXXXXXX"},            # wall mini
            {39: "XXXX..XXXX", 38: "XXX..XXXXX", 37: "XXXX.XXXXX"},            # S/Z spin (s2 all-spin)
            {39: "XXXXXXX...", 38: "XXXXXXXXX.", 37: "XXXXXXXX.."},            # L/J spin
        ]
        for k, spec in enumerate(tslots):
            for piece, held in (("T", "I"), ("S", "T"), ("L", "J"), ("Z", "T")):
                g, rng = base(ruleset, 300 + k)
                pl = g.players[0]
                fill(pl, spec)
                pl.piece = Piece(type=piece); pl.piece.move_to_spawn()
                pl.held_piece = held
                pl.stats.b2b = (k % 3) * 3 - 1
                pl.stats.b2b_level = k % 3
                pl.stats.combo = k % 4
                pl.garbage_to_receive = [3] * (k % 4) + [7] * (k % 2)
                g.players[1].garbage_to_receive = [1, 1, 9][: k % 4]
                out.append((g, rng, ruleset))
        # tall stacks: spawn-blocked losses, hold fallbacks with an empty queue
        for k, height in enumerate((18, 19, 20, 21, 22)):
            g, rng = base(ruleset, 400 + k)
            pl = g.players[0]
            for r in range(40 - height, 40):
                for c in range(10):
                    pl.board.grid[r][c] = 0 if c == (r * 3 + k) % 10 else 1
            pl.garbage_to_receive = [k, k, (k + 5) % 10]
            out.append((g, rng, ruleset))
        g, rng = base(ruleset, 450)
        pl = g.players[0]
        pl.queue.pieces = []                      # search-time state: no previews left
        pl.held_piece = "J"
        out.append((g, rng, ruleset))
        g, rng = base(ruleset, 451)
        pl = g.players[0]
        pl.queue.pieces = []
        pl.held_piece = None                      # last piece: afterwards no_move
        out.append((g, rng, ruleset))
        # player 1 to move with garbage pending on both sides
        g, rng = base(ruleset, 452)
        g.turn = 1
        fill(g.players[1], {39: "XXXXXXXXX.", 38: "XXXXXXXXX.", 37: "XXXXXXXX.."})
        g.players[1].garbage_to_receive = [2, 2, 2, 2, 5, 5]
        g.players[0].garbage_to_receive = [8]
        g.players[1].piece = Piece(type="I"); g.players[1].piece.move_to_spawn()
        out.append((g, rng, ruleset))
    return out



def cave_states(n_keep=48, n_try=6000):
    """Random cave-like stacks with a T piece, kept when the reference generator finds a
    placement that used the last kick (planes 23-26) -- the order-dependent case H1 of
    SURVEY.md section 7 -- or, for a few, any spin plane."""
    rs = np.random.RandomState(2026)
    kept, kept_spin = [], 0
    for t in range(n_try):
        ruleset = "s2" if t % 2 else "s1"
        g, rng = new_game(ruleset, 500 + t)
        pl = g.players[0]
        h = rs.randint(3, 9)
        dens = rs.uniform(0.55, 0.9)
        for r in range(40 - h, 40):
            row = (rs.rand(10) < dens)
            if row.all():
                row[rs.randint(10)] = False
            for c in range(10):
                pl.board.grid[r][c] = 1 if row[c] else 0
        pl.piece = Piece(type="T"); pl.piece.move_to_spawn()
        pl.held_piece = MINOS[rs.randint(7)]
        pl.stats.b2b = int(rs.randint(-1, 6)); pl.stats.b2b_level = int(rs.randint(0, 3)); pl.stats.combo = int(rs.randint(0, 4))
        moves = legal_moves(g)
        planes = [(m >> 10) & 31 for m in moves]
        if any(p >= 23 for p in planes):
            kept.append((g, rng, ruleset))
        elif kept_spin < 16 and any(p >= 19 for p in planes):
            kept.append((g, rng, ruleset)); kept_spin += 1
        if len(kept) >= n_keep:
            break
    return kept

def gen_step_cases(sample_states):
    states_in, case_state, case_move, states_out = [], [], [], []
    srcs = []
    for g, rng, ruleset in crafted_states() + cave_states():
        srcs.append(rh.pack_game(g, rng, 0, ruleset))
    rs = np.random.RandomState(11)
    idx = rs.choice(len(sample_states), size=40, replace=False)
    srcs += [sample_states[i] for i in sorted(idx)]
    tsp = [0, 0, 0]
    for buf in srcs:
        g, rng, ply = game_from_state(buf)
        moves = legal_moves(g)
        sid = len(states_in)
        states_in.append(buf)
        ruleset = g.ruleset
        for m in moves:
            g2, rng2, _ = game_from_state(buf)
            plane, col, row = decode(m)
            g2.make_move((plane, col, row), add_bag=True, add_history=False)
            case_state.append(sid); case_move.append(m)
            states_out.append(rh.pack_game(g2, rng2, ply + (0 if g2.turn == g.turn else 1), ruleset))
            if plane >= 23: tsp[2] += 1
            elif plane >= 19: tsp[1] += 1
            else: tsp[0] += 1
    np.savez_compressed(
        os.path.join(OUT, "step_cases.npz"),
        seed=np.uint64(SEED),
        states_in=np.frombuffer(b"".join(states_in), dtype=np.uint8).reshape(-1, 320),
        case_state=np.array(case_state, dtype=np.int32), case_move=np.array(case_move, dtype=np.uint16),
        states_out=np.frombuffer(b"".join(states_out), dtype=np.uint8).reshape(-1, 320))
    print("step_cases: %d states, %d (state, move) cases, planes plain/spin/spin+lastkick = %s" %
          (len(states_in), len(case_move), tsp))


# ---------------------------------------------------------------------------------------------
def gen_attack():
    rows = []
    for rs_id, ruleset in ((1, "s1"), (2, "s2")):
        for rows_cleared in range(5):
            for tspin in (0, 1):
                for mini in (0, 1):
                    for ac in (0, 1):
                        for combo in range(0, 21):
                            for b2b in list(range(-1, 31)) + [66, 67, 184, 185, 503, 504, 1369, 1370]:
                                for lvl in range(0, 9):
                                    st = Stats(ruleset)
                                    st.combo, st.b2b, st.b2b_level = combo, b2b, lvl
                                    a = st.get_attack(rows_cleared, bool(tspin), bool(mini), bool(ac), "T")
                                    rows.append((rs_id, rows_cleared, tspin, mini, ac, combo, b2b, lvl,
                                                 a, st.combo, st.b2b, st.b2b_level))
    arr = np.array(rows, dtype=np.int16)
    np.savez_compressed(os.path.join(OUT, "attack.npz"), table=arr)
    print("attack: %d rows" % len(rows))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    states = gen_env_traces()
    gen_movegen(states)
    gen_step_cases(states)
    gen_attack()
