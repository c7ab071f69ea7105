"""Gating match (battle_networks, ai.py:1975-2114) on the device against the oracle: two different
deterministic evaluators alternate colours over 12 concurrent games; every game's move sequence and
winner must equal the oracle's statement of aplay_battle_game, and the win tally follows."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle_py as op
import tetris_rl_amd as trl
from tetris_rl_amd import gating

pytestmark = pytest.mark.gpu
SALT1, SALT2 = 0x1111111111111111, 0x2222222222222222


def test_battle_matches_oracle():
    games, seed = 12, 99
    cfg = trl.Config(visual=False, model="pytorch", MAX_ITER=16, training=False)
    L = trl.load_library()
    L.trl_debug_stub_eval_salted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    holder = {}

    def make_eval(salt):
        def ev(grids, vec):
            b = holder["b"]
            v = torch.empty(b.n, dtype=torch.float32, device=b.device)
            p = torch.empty((b.n, 11583), dtype=torch.float32, device=b.device)
            assert L.trl_debug_stub_eval_salted(b.engine._h, v.data_ptr(), p.data_ptr(), salt) == 0
            return v, p
        return ev

    b = gating.Battle(cfg, make_eval(SALT1), make_eval(SALT2), games, seed=seed)
    holder["b"] = b
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
    wins, accepted = gating.battle_networks(a, cfg, b, cfg, 0.52, "moreorequal", 6)
    assert wins.sum() == 6 and accepted in (True, False, None)
    assert accepted == gating._check_threshold(wins, 6, 0.52, "moreorequal")
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
