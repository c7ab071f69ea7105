"""The CPU oracle (oracle/tetris_oracle.c) against the fixtures the Python reference produced
(oracle/gen_golden.py).  This is what pins the oracle; the HIP engine is then compared with
the oracle (tests/test_hip_parity.py) and with the same fixtures."""
import os

import numpy as np
import pytest

import oracle_py as op


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_attack_table_exhaustive(golden_dir):
    """Stats.get_attack (stats.py:45-157): 604 800 (ruleset, rows, tspin, mini, all-clear, combo,
    b2b, level) combinations -> attack and the updated combo / b2b / b2b_level."""
    t = _load(golden_dir, "attack.npz")["table"]
    out = op.attack_table(t[:, :8])
    bad = np.nonzero((out != t[:, 8:12]).any(axis=1))[0]
    assert bad.size == 0, "first mismatch: in=%s ref=%s oracle=%s" % (t[bad[0], :8], t[bad[0], 8:], out[bad[0]])


def _iter_games(d):
    counts = d["legal_counts"]
    offs = np.concatenate([[0], np.cumsum(counts)])
    moves_off = 0
    for ruleset, stream, driver, n_plies, s_off, l_off in d["meta"]:
        yield dict(ruleset=int(ruleset), stream=int(stream), driver=int(driver), n_plies=int(n_plies),
                   states=d["states"][s_off:s_off + n_plies + 1],
                   moves=d["moves"][moves_off:moves_off + n_plies],
                   legal=[d["legal_flat"][offs[l_off + k]:offs[l_off + k + 1]] for k in range(n_plies + 1)])
        moves_off += n_plies


def test_env_traces_replay(golden_dir):
    """Every ply of 80 reference games: legal move list (get_move_matrix 'convolutional' in
    argwhere order) and the full two-player state after Game.make_move."""
    d = _load(golden_dir, "env_traces.npz")
    seed = int(d["seed"])
    n_states = 0
    for gm in _iter_games(d):
        g = op.OracleGame(ruleset=gm["ruleset"], seed=seed, stream=gm["stream"])
        for k in range(gm["n_plies"] + 1):
            st, ref = g.export(), gm["states"][k].copy()
            st[304:308] = 0; ref[304:308] = 0    # pol_ctr: the replay applies recorded moves, it draws none
            assert np.array_equal(st, ref), "state mismatch game %s ply %d" % (gm["stream"], k)
            lm = g.legal_moves()
            assert np.array_equal(lm, gm["legal"][k]), "legal-move mismatch game %s ply %d" % (gm["stream"], k)
            n_states += 1
            if k < gm["n_plies"]:
                g.make_move(gm["moves"][k])
    assert n_states == d["states"].shape[0]


def test_uniform_games_reproduce_from_seed_only(golden_dir):
    """The uniform-policy games need no move list: the oracle's rollout (same RNG definition
    as the engine) must reproduce the reference's states from (seed, stream) alone."""
    d = _load(golden_dir, "env_traces.npz")
    seed = int(d["seed"])
    n = 0
    for gm in _iter_games(d):
        if gm["driver"] != 0:
            continue
        g = op.OracleGame(ruleset=gm["ruleset"], seed=seed, stream=gm["stream"])
        for k in range(1, gm["n_plies"] + 1):
            last = (k == gm["n_plies"])
            if last:
                break   # the last ply ends the game: the rollout would auto-reset the slot
            g.rollout_random(1)
            assert np.array_equal(g.export(), gm["states"][k])
        n += 1
    assert n == 48


def test_step_cases(golden_dir):
    """(state, move) -> state for every legal move of crafted / cave / sampled positions:
    T-spin planes with and without the last-kick flag, all-clears, surge, garbage cancel and
    spawn, spawn-blocked losses, empty-queue hold fallback."""
    d = _load(golden_dir, "step_cases.npz")
    seed = int(d["seed"])
    sin, cs, cm, sout = d["states_in"], d["case_state"], d["case_move"], d["states_out"]
    # the legal list of each input state must be exactly the moves the reference enumerated
    for sid in range(sin.shape[0]):
        g = op.OracleGame(state=sin[sid], seed=seed)
        assert np.array_equal(g.legal_moves(), cm[cs == sid]), "legal list of crafted state %d" % sid
    for i in range(cm.shape[0]):
        g = op.OracleGame(state=sin[cs[i]], seed=seed)
        g.make_move(cm[i])
        assert np.array_equal(g.export(), sout[i]), "case %d (state %d, move %d)" % (i, cs[i], cm[i])


def _state_for_case(rows, piece, held, q0):
    st = np.zeros(320, dtype=np.uint8)
    st[0:80] = np.asarray(rows, dtype="<u2").view(np.uint8)
    st[112] = q0 & 15                 # queue nibble 0
    st[132:134] = np.array([-1], dtype="<i2").view(np.uint8)   # b2b
    st[137], st[138] = piece, held
    st[139] = 1 if q0 != 7 else 0     # queue_len
    st[144 + 132:144 + 134] = np.array([-1], dtype="<i2").view(np.uint8)
    st[144 + 137] = st[144 + 138] = 7
    st[314] = 1
    return st


def test_movegen_reference_boards_and_samples(golden_dir):
    """get_move_matrix on the reference's own test boards (util.py:34-37 padded to 40 rows,
    tests.py:13-58) x 7 pieces x {held None, held == piece}, the empty board, and 592 sampled
    positions; 'convolutional' everywhere, 'brute-force' where the fixture has it."""
    d = _load(golden_dir, "movegen.npz")
    cc = np.concatenate([[0], np.cumsum(d["conv_counts"])])
    bc = np.concatenate([[0], np.cumsum(d["brute_counts"])])
    n = d["rows"].shape[0]
    for i in range(n):
        st = _state_for_case(d["rows"][i], int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i]))
        conv = op.mask_to_moves(op.move_matrix_from_state(st, 0, 0))
        assert np.array_equal(conv, d["conv_flat"][cc[i]:cc[i + 1]]), "conv case %d" % i
        if d["has_brute"][i]:
            brute = op.mask_to_moves(op.move_matrix_from_state(st, 0, 1))
            assert np.array_equal(brute, d["brute_flat"][bc[i]:bc[i + 1]]), "brute case %d" % i


def test_movegen_t_piece_disagreements(golden_dir):
    """H1 at depth: 240 T-piece positions where the reference's 'convolutional' and 'brute-force' lists differ (+120
    where they agree); the oracle reproduces both lists on every one"""
    d = _load(golden_dir, "movegen_t.npz")
    cc = np.concatenate([[0], np.cumsum(d["conv_counts"])])
    bc = np.concatenate([[0], np.cumsum(d["brute_counts"])])
    assert int(d["differ"].sum()) >= 200
    for i in range(d["rows"].shape[0]):
        st = _state_for_case(d["rows"][i], int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i]))
        conv = op.mask_to_moves(op.move_matrix_from_state(st, 0, 0))
        brute = op.mask_to_moves(op.move_matrix_from_state(st, 0, 1))
        assert np.array_equal(conv, d["conv_flat"][cc[i]:cc[i + 1]]), "conv case %d" % i
        assert np.array_equal(brute, d["brute_flat"][bc[i]:bc[i + 1]]), "brute case %d" % i
        assert bool(d["differ"][i]) == (not np.array_equal(conv, brute))


def test_reference_known_answers(golden_dir):
    """tests.py:23-32: O piece, held 'O', empty board -> exactly 9 placements (both algorithms);
    tests.py:45-58: convolutional == brute-force on the four reference boards."""
    empty = np.zeros(40, dtype=np.uint16)
    st = _state_for_case(empty, 2, 2, 0)
    assert int(op.move_matrix_from_state(st, 0, 1).sum()) == 9
    assert int(op.move_matrix_from_state(st, 0, 0).sum()) == 9
    d = _load(golden_dir, "movegen.npz")
    for i in range(int(d["n_reference_cases"])):
        st = _state_for_case(d["rows"][i], int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i]))
        a = op.move_matrix_from_state(st, 0, 0)
        b = op.move_matrix_from_state(st, 0, 1)
        assert np.array_equal(a, b)
        assert 1 <= int(b.sum()) <= b.size
