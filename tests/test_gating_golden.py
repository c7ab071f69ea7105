"""The gating match pinned on the reference (tests/golden/battle.npz, written by oracle/gen_golden_battle.py from the
reference's _battle_networks_async / aplay_battle_game / amcts and from its sequential battle_networks loop with two
salted stub evaluators): the oracle's to_play_battle_stub reproduces every game (moves, winner), and the host-side
tallies of tetris_rl_amd.gating reproduce the reference's (wins, accepted) in both forms, early termination included."""
import os

import numpy as np

import oracle_py as op
from tetris_rl_amd import gating


def _fixture(golden_dir):
    d = np.load(os.path.join(golden_dir, "battle.npz"))
    off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    return d, off


def test_oracle_battle_games_match_reference(golden_dir):
    d, off = _fixture(golden_dir)
    cfg = op.default_mcts_config(max_iter=int(d["max_iter"]), training=0)
    for i in range(int(d["games"])):
        moves, winner = op.play_battle_stub(2, int(d["seed"]), int(d["stream0"]) + i, cfg, i % 2, int(d["salt1"]), int(d["salt2"]))
        assert moves.tolist() == d["game_moves"][off[i]:off[i + 1]].tolist(), "game %d" % i
        assert winner == d["game_winner"][i]


def test_tallies_match_reference(golden_dir):
    d, _ = _fixture(golden_dir)
    winners = [int(w) for w in d["game_winner"]]
    games = int(d["games"])
    wins = gating.tally_concurrent(winners)
    assert np.array_equal(wins, d["async_wins"]) and wins.dtype == np.float64
    acc = gating._check_threshold(wins, games, float(d["async_threshold"]), "moreorequal")
    assert (-1 if acc is None else int(acc)) == int(d["async_accepted"])
    assert len(set(d["seq_played"].tolist())) >= 3                  # the fixture does stop early at different points
    for i in range(len(d["seq_threshold"])):
        thr = float(d["seq_threshold"][i])
        w = np.zeros(2, dtype=int)
        n, acc = gating.scan_sequential(w, winners, 0, games, None if thr < 0 else thr, str(d["seq_type"][i]))
        assert n == d["seq_played"][i] and np.array_equal(w, d["seq_wins"][i])
        assert (-1 if acc is None else int(acc)) == d["seq_accepted"][i]
    # the same scan fed wave by wave (as battle_networks does on the device) stops at the same game
    for wave in (1, 3, 8):
        w, i, acc = np.zeros(2, dtype=int), 0, None
        while i < games and acc is None:
            n, acc = gating.scan_sequential(w, winners[i:i + wave], i, games, 0.55, "moreorequal")
            i += n
        assert i == 7 and acc is False and w.tolist() == [3, 4]


def test_draw_in_the_sequential_form_counts_half_a_point():
    """ai.py:1991 / 2022: the reference keeps integer win counters and does `wins += 0.5` for a draw, which numpy refuses
    (UFuncTypeError) -- inside self_play_loop that would abort the loop after the set was generated and the challenger
    trained.  battle_networks here scans with a float array: a draw is half a point each, and the result is the
    reference's integer array whenever no game was drawn."""
    w = np.zeros(2, dtype=float)
    n, acc = gating.scan_sequential(w, [-1, 0, 1, -1], 0, 8, None, "more")
    assert n == 4 and acc is None and w.tolist() == [1.0, 3.0]          # game 1: colours flipped -> network 2 wins it
    assert gating._as_reference_wins(w).dtype.kind == "i"
    w = np.zeros(2, dtype=float)
    gating.scan_sequential(w, [-1, 0], 0, 8, None, "more")
    assert gating._as_reference_wins(w).tolist() == [0.5, 1.5]
    assert gating.tally_concurrent([-1, 0, 1]).tolist() == [0.5, 2.5]   # the concurrent form counts halves (game 1: colours flipped)
