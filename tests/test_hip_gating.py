"""Gating match (battle_networks, ai.py:1975-2114) on the device against the reference's fixture and the oracle: two
different deterministic evaluators alternate colours over concurrent games; every game's move sequence and winner must
equal the reference's (tests/golden/battle.npz) / the oracle's statement of aplay_battle_game, and the tallies follow."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_py as op
import tetris_rl_amd as trl
from tetris_rl_amd import gating

pytestmark = pytest.mark.gpu
SALT1, SALT2 = 0x1111111111111111, 0x2222222222222222


class StubEval:
    """the on-device stub evaluator with a salt ("a network"); Battle binds it to its engine"""

    def __init__(self, salt):
        self.salt, self.b = salt, None
        self.L = trl.load_library()
        self.L.trl_debug_stub_eval_salted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]

    def bind(self, battle):
        self.b = battle

    def __call__(self, grids, vec):
        b = self.b
        v = torch.empty(b.n, dtype=torch.float32, device=b.device)
        p = torch.empty((b.n, 11583), dtype=torch.float32, device=b.device)
        assert self.L.trl_debug_stub_eval_salted(b.engine._h, v.data_ptr(), p.data_ptr(), self.salt) == 0
        return v, p


def test_battle_matches_reference_fixture(golden_dir):
    """tests/golden/battle.npz: 8 games of the reference's battle_networks between two salted stub evaluators.  The device
    Battle plays the same games move for move; battle_networks returns the reference's (wins, accepted) in the concurrent
    form (batched_inference, ai.py:1981-1988) and in the sequential form with early termination (ai.py:1996-2049),
    whatever the wave size."""
    import os
    d = np.load(os.path.join(golden_dir, "battle.npz"))
    off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    games, seed, s1, s2 = int(d["games"]), int(d["seed"]), int(d["salt1"]), int(d["salt2"])
    cfg = trl.Config(visual=False, model="pytorch", MAX_ITER=int(d["max_iter"]), training=False, batched_inference=True)
    b = gating.Battle(cfg, StubEval(s1), StubEval(s2), games, seed=seed, index_base=int(d["stream0"]))
    wins = b.run()
    for g in range(games):
        assert [m for _, m in sorted(b.moves[g])] == d["game_moves"][off[g]:off[g + 1]].tolist(), "game %d" % g
        assert b.winners[g] == d["game_winner"][g]
    assert int(d["stream0"]) % 2 == 0 and np.array_equal(wins, d["async_wins"])
    # battle_networks plays game i on stream i: use the fixture's games by seeding the streams through index_base --
    # here through a shifted match: the fixture's streams start at stream0, so compare on winners replayed from it
    cfg_seq = cfg.copy(); cfg_seq.batched_inference = False
    real_battle = gating.Battle

    def shifted(config, e1, e2, n, seed=0, device=0, streams=None, nets=None, index_base=0):
        return real_battle(config, e1, e2, n, seed=seed, device=device, nets=nets, index_base=index_base + int(d["stream0"]))

    gating.Battle = shifted
    try:
        w, acc = gating.battle_networks(None, cfg, None, cfg, float(d["async_threshold"]), "moreorequal", games, seed=seed,
                                        evaluators=(StubEval(s1), StubEval(s2)))
        assert np.array_equal(w, d["async_wins"]) and (-1 if acc is None else int(acc)) == int(d["async_accepted"])
        for i in range(len(d["seq_threshold"])):
            thr = float(d["seq_threshold"][i])
            for wave in (3, 64):
                w, acc = gating.battle_networks(None, cfg_seq, None, cfg_seq, None if thr < 0 else thr, str(d["seq_type"][i]),
                                                games, seed=seed, evaluators=(StubEval(s1), StubEval(s2)), wave=wave)
                assert w.dtype.kind == "i" and np.array_equal(w, d["seq_wins"][i]), (i, wave)
                assert (-1 if acc is None else int(acc)) == d["seq_accepted"][i]
    finally:
        gating.Battle = real_battle


def test_battle_matches_oracle():
    games, seed = 12, 99
    cfg = trl.Config(visual=False, model="pytorch", MAX_ITER=16, training=False)
    b = gating.Battle(cfg, StubEval(SALT1), StubEval(SALT2), games, seed=seed)
    wins = b.run()
    want = np.zeros(2)
    ocfg = op.default_mcts_config(max_iter=16, training=0)
    for g in range(games):
        side = g % 2
        moves, winner = op.play_battle_stub(2, seed, g, ocfg, side, SALT1, SALT2)
        mine = [m for _, m in sorted(b.moves[g])]
        assert mine == moves.tolist(), "game %d" % g
        if winner == -1:
            want += 0.5
        elif side == 0:
            want[winner] += 1
        else:
            want[1 - winner] += 1
    assert np.array_equal(wins, want) and wins.sum() == games


def test_battle_networks_signature_and_threshold():
    torch.manual_seed(1)
    a = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=8))
    b = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=8))
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=2, filters=8), MAX_ITER=8)
    cfg.batched_inference = True                      # the concurrent form: every game is played, threshold afterwards
    wins, accepted = gating.battle_networks(a, cfg, b, cfg, 0.52, "moreorequal", 6, seed=0)
    assert wins.sum() == 6 and accepted in (True, False, None)
    assert accepted == gating._check_threshold(wins, 6, 0.52, "moreorequal")
    cfg.batched_inference = False                     # the sequential form stops at the deciding game
    wins2, accepted2 = gating.battle_networks(a, cfg, b, cfg, 0.52, "moreorequal", 6, seed=0)
    assert wins2.dtype.kind == "i" and wins2.sum() <= 6 and accepted2 == accepted
    if accepted2 is not None:
        assert gating._check_threshold(wins2, 6, 0.52, "moreorequal") == accepted2
    with pytest.raises(NotImplementedError):
        gating.battle_networks(a, cfg, b, trl.Config(ruleset="s1"), 0.5, "more", 2)
    assert gating._check_threshold(np.array([6.0, 4.0]), 10, 0.55, "more") is True
    assert gating._check_threshold(np.array([4.0, 6.0]), 10, 0.55, "moreorequal") is False
    assert gating._check_threshold(np.array([5.0, 4.0]), 10, 0.55, "more") is None


def test_native_two_network_tick():
    """battle_networks with two AlphaSame nets on the fused tower: one tower per leaf and one expand launch per
    network (trl_mcts_expand_alphasame_masked) -- deterministic, every game decided, and the same first moves as
    the generic path that runs both nets' full forward pass over every slot (float32 rounding of the heads aside)"""
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=16, training=False)
    torch.manual_seed(3); a = trl.AlphaSame(mc).cuda().prepare()
    torch.manual_seed(4); b = trl.AlphaSame(mc).cuda().prepare()
    games = 32
    runs = []
    for native in (True, True, False):
        bt = gating.Battle(cfg, a.evaluate_packed, b.evaluate_packed, games, seed=9, nets=(a, b) if native else None)
        assert (bt.nets is not None) == native
        wins = bt.run()
        assert wins.sum() == games
        runs.append((wins, {g: [m for _, m in sorted(v)] for g, v in bt.moves.items()}))
    assert np.array_equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    same_first = sum(runs[0][1][g][0] == runs[2][1][g][0] for g in range(games))
    assert same_first >= games - 2


def test_gating_and_single_searches_never_open_with_random_moves():
    """Config.use_random_starting_moves is play_game's (ai.py:1588-1608): battle_networks and MCTS() search from ply 0
    (ai.py:2012-2014).  With the flag on, no record of a match is a random move and every first move has a full search
    behind it."""
    cfg = trl.Config(visual=False, model="pytorch", MAX_ITER=12, training=False, use_random_starting_moves=True)
    b = gating.Battle(cfg, StubEval(SALT1), StubEval(SALT2), 6, seed=5)
    e = b.engine
    b.tick(12)
    b.stream.synchronize()
    r = e.mcts_drain_records()
    r = r[r["kind"] == trl.REC_MOVE]
    assert len(r) == 6 and not r["random_move"].any()
    nch = r["n_children"].astype(np.int64)
    assert all(int(m["visits"][:k].sum()) == 11 for m, k in zip(r, nch))          # 12 simulations, not a one-simulation pick
    # the same games as with the flag off
    cfg0 = cfg.copy(); cfg0.use_random_starting_moves = False
    b0 = gating.Battle(cfg0, StubEval(SALT1), StubEval(SALT2), 6, seed=5)
    b0.tick(12)
    b0.stream.synchronize()
    r0 = b0.engine.mcts_drain_records()
    r0 = r0[r0["kind"] == trl.REC_MOVE]
    assert np.array_equal(np.sort(r["chosen"]), np.sort(r0["chosen"]))
