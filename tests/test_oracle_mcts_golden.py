"""The oracle's search (oracle/tetris_oracle_mcts.c) against the reference's MCTS run with the
stub evaluator (oracle/gen_golden_mcts.py): root children, visit counts, Q, priors, chosen move,
tree size, and the running garbage-draw counter after the search."""
import os

import numpy as np
import pytest

import oracle_py as op

VARIANTS = {
    "eval": dict(training=0),
    "train": dict(training=1),
    "train_nosoftmax": dict(training=1, use_root_softmax=0),
    "eval_absfpu": dict(training=0, fpu_reduction=0, fpu_value=0.4),
    "train_forced": dict(training=1, use_forced=1),
}


def test_stub_evaluator_matches_harness():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import importlib
    g = op.OracleGame(ruleset=2, seed=1, stream=5)
    g.rollout_random(7)
    st = g.export()
    v, pol = op.stub_eval_state(st)
    # independent numpy statement of the stub (the one the fixtures were generated with)
    gm = importlib.import_module("gen_golden_mcts")
    v2, pol2 = gm.stub_eval_state(bytes(st))
    assert v == v2 and np.array_equal(pol, pol2)
    assert (pol == 0).mean() > 0.03 and 0 <= v < 1


def test_mcts_single_searches(golden_dir):
    d = np.load(os.path.join(golden_dir, "mcts.npz"))
    seed = int(d["seed"])
    names = [str(x) for x in d["variant_names"]]
    off = np.concatenate([[0], np.cumsum(d["child_counts"])])
    for i in range(d["states"].shape[0]):
        cfg = op.default_mcts_config(**VARIANTS[names[d["variant"][i]]])
        g = op.OracleGame(state=d["states"][i], seed=seed)
        r = op.mcts_stub(g, cfg)
        sl = slice(off[i], off[i + 1])
        tag = "case %d (%s)" % (i, names[d["variant"][i]])
        assert np.array_equal(r["moves"], d["child_moves"][sl]), tag
        assert np.array_equal(r["visits"], d["child_visits"][sl]), tag
        assert r["move"] == d["move"][i] and r["save"] == d["save"][i], tag
        assert r["n_nodes"] == d["n_nodes"][i] and r["root_visit"] == d["root_visit"][i], tag
        assert np.array_equal(r["q"], d["child_q"][sl]), tag          # fp64 bit patterns
        assert np.array_equal(r["p"], d["child_p"][sl]), tag
        assert int(g.export()[296:300].view("<u4")[0]) == d["garb_ctr"][i], tag


def test_play_game_matches_reference(golden_dir):
    """play_game (ai.py:1570-1699) with the stub evaluator: every ply's pre-search state, chosen
    move, save flag and root (move, visits) vector of three reference games."""
    d = np.load(os.path.join(golden_dir, "selfplay.npz"))
    seed = int(d["seed"])
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    for gi in range(d["game_ruleset"].shape[0]):
        cfg = op.default_mcts_config(max_iter=int(d["max_iter"]), training=int(d["game_training"][gi]))
        r = op.play_game_stub(int(d["game_ruleset"][gi]), seed, int(d["game_stream"][gi]), cfg)
        assert r["plies"] == d["game_plies"][gi]
        sl = slice(p_off[gi], p_off[gi + 1])
        a, b = r["ply_state"].copy(), d["ply_state"][sl].copy()
        a[:, 304:308] = 0; b[:, 304:308] = 0
        assert np.array_equal(a, b), "pre-search states, game %d" % gi
        assert np.array_equal(r["ply_move"], d["ply_move"][sl])
        assert np.array_equal(r["ply_save"], d["ply_save"][sl])
        assert np.array_equal(r["ply_nchild"], d["ply_child_counts"][sl])
        csl = slice(c_off[p_off[gi]], c_off[p_off[gi + 1]])
        assert np.array_equal(r["child_moves"], d["ply_child_moves"][csl])
        assert np.array_equal(r["child_visits"], d["ply_child_visits"][csl])


def _cfg160(d, gi):
    return op.default_mcts_config(max_iter=int(d["game_max_iter"][gi]), training=int(d["game_training"][gi]),
                                  use_tanh=int(d["game_use_tanh"][gi]), random_start=int(d["game_random_start"][gi]),
                                  use_root_softmax=int(d["game_root_softmax"][gi]))


def test_play_game_160_tanh_and_random_start(golden_dir):
    """whole reference games at MAX_ITER=160 with training=True (C1 / C3's game loop), a use_tanh game and two
    use_random_starting_moves games (oracle/gen_golden_selfplay160.py): per-ply state, move, save flag, root vector"""
    d = np.load(os.path.join(golden_dir, "selfplay160.npz"))
    seed = int(d["seed"])
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    assert d["game_n_rand"].sum() >= 2
    for gi in range(d["game_ruleset"].shape[0]):
        r = op.play_game_stub(int(d["game_ruleset"][gi]), seed, int(d["game_stream"][gi]), _cfg160(d, gi))
        tag = str(d["game_tag"][gi])
        assert r["plies"] == d["game_plies"][gi], tag
        sl = slice(p_off[gi], p_off[gi + 1])
        a, b = r["ply_state"].copy(), d["ply_state"][sl].copy()
        a[:, 304:308] = 0; b[:, 304:308] = 0
        assert np.array_equal(a, b), "pre-search states, " + tag
        assert np.array_equal(r["ply_move"], d["ply_move"][sl]), tag
        assert np.array_equal(r["ply_save"], d["ply_save"][sl]), tag
        assert np.array_equal(r["ply_nchild"], d["ply_child_counts"][sl]), tag
        csl = slice(c_off[p_off[gi]], c_off[p_off[gi + 1]])
        assert np.array_equal(r["child_moves"], d["ply_child_moves"][csl]), tag
        assert np.array_equal(r["child_visits"], d["ply_child_visits"][csl]), tag


def test_mcts_tanh_searches(golden_dir):
    d = np.load(os.path.join(golden_dir, "selfplay160.npz"))
    off = np.concatenate([[0], np.cumsum(d["tanh_child_counts"])])
    for i in range(d["tanh_states"].shape[0]):
        cfg = op.default_mcts_config(training=int(d["tanh_variant"][i]), use_tanh=1)
        r = op.mcts_stub(op.OracleGame(state=d["tanh_states"][i], seed=int(d["seed"])), cfg)
        sl = slice(off[i], off[i + 1])
        assert np.array_equal(r["moves"], d["tanh_child_moves"][sl]) and np.array_equal(r["visits"], d["tanh_child_visits"][sl])
        assert r["move"] == d["tanh_move"][i] and r["save"] == d["tanh_save"][i]
        assert np.array_equal(r["q"], d["tanh_child_q"][sl]) and np.array_equal(r["p"], d["tanh_child_p"][sl])
        assert d["tanh_child_q"][sl].min() < 0                     # the negative half of the value range is exercised


def test_searches_with_the_brute_force_generator(golden_dir):
    """Config.move_algorithm (ai.py:84, 414): reference searches with 'brute-force' and with 'convolutional' on T-piece
    positions where the two generators disagree at the root (tests/golden/mcts_bf.npz, oracle/gen_golden_mcts_bf.py)"""
    d = np.load(os.path.join(golden_dir, "mcts_bf.npz"))
    off = np.concatenate([[0], np.cumsum(d["child_counts"])])
    differ = 0
    for i in range(len(d["states"])):
        cfg = op.default_mcts_config(max_iter=int(d["max_iter"]), training=0, move_algorithm=int(d["algo"][i]))
        r = op.mcts_stub(op.OracleGame(state=d["states"][i], seed=int(d["seed"])), cfg)
        sl = slice(off[i], off[i + 1])
        assert np.array_equal(r["moves"], d["child_moves"][sl]) and np.array_equal(r["visits"], d["child_visits"][sl]), i
        assert r["move"] == d["move"][i] and np.array_equal(r["q"], d["child_q"][sl]) and np.array_equal(r["p"], d["child_p"][sl]), i
        if i % 2 == 1:
            prev = slice(off[i - 1], off[i])
            differ += not np.array_equal(d["child_moves"][prev], d["child_moves"][sl])
    assert differ >= 10                                  # the fixture does tell the two generators apart
