"""Single-game object API (Game / get_move_matrix / evaluate / MCTS with the reference's signatures)
on the device, checked against the oracle."""
import numpy as np
import pytest
import torch

import oracle_py as op
import tetris_rl_amd as trl

pytestmark = pytest.mark.gpu


def test_game_object_follows_oracle():
    g = trl.Game("s2", seed=77, stream=3)
    g.setup()
    o = op.OracleGame(ruleset=2, seed=77, stream=3)
    for ply in range(30):
        assert np.array_equal(g._pack()[:288], o.export()[:288])
        mask = trl.get_move_matrix(g.players[g.turn], "convolutional")
        legal = o.legal_moves()
        assert np.array_equal(op.mask_to_moves(mask), legal)
        m = int(legal[(ply * 7) % len(legal)])
        g.make_move(trl.decode_move(m))
        o.make_move(m)
        if g.is_terminal:
            break
    assert g.turn == int(o.export()[312]) and g.winner == o.winner()
    c = g.copy()
    assert np.array_equal(c._pack(), g._pack())


def _player_on_board(rows, piece_letter, held=None):
    """tests.py:4-12 / 36-43: a fresh s1 game whose side to move gets `rows` as its board and `piece_letter` at spawn"""
    g = trl.Game("s1", seed=7, stream=7)
    g.setup()
    pl = g.players[g.turn]
    for r in range(40):
        for c in range(10):
            pl.board.grid[r][c] = 1 if (int(rows[r]) >> c) & 1 else 0
    pl.piece = trl.compat.Piece(piece_letter)
    pl.held_piece = held
    return pl


def test_reference_own_movegen_tests_through_compat(golden_dir):
    """The reference's own tests of this path (tests.py:13-58), calls as the reference writes them: get_move_matrix with
    its DEFAULT argument (= 'brute-force') finds 1..capacity moves on the four reference boards x 7 pieces, exactly 9 for
    O on the empty board with O held, and algo='convolutional' gives the identical matrix on those boards."""
    import os
    d = np.load(os.path.join(golden_dir, "movegen.npz"))
    boards = [d["rows"][i] for i in range(0, int(d["n_reference_cases"]) - 14, 14)]          # 4 boards x 7 pieces x 2 holds
    assert len(boards) == 4
    for rows in boards:
        for letter in trl.compat.MINOS:
            pl = _player_on_board(rows, letter)
            bf = trl.get_move_matrix(pl)                                       # default argument
            assert bf.shape == (27, 39, 11) and bf.dtype == bool and 1 <= int(bf.sum()) <= bf.size
            assert np.array_equal(bf, trl.get_move_matrix(pl, algo="brute-force"))
            assert np.array_equal(bf, trl.get_move_matrix(pl, algo="convolutional"))
    pl = _player_on_board(np.zeros(40, dtype=np.uint16), "O", held="O")
    assert int(trl.get_move_matrix(pl).sum()) == 9


def test_unknown_algorithm_and_illegal_move():
    g = trl.Game("s1", seed=1)
    g.setup()
    with pytest.raises(ValueError):
        trl.get_move_matrix(g.players[0], "harddrop")
    with pytest.raises(ValueError):
        g.make_move((26, 0, 0))


def test_mcts_and_evaluate_signatures():
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=8)).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=2, filters=8), MAX_ITER=24)
    g = trl.Game("s2", seed=5)
    g.setup()
    before = g._pack().copy()
    value, policy = trl.evaluate(cfg, g, net)
    assert 0.0 <= value <= 1.0 and policy.shape == (27, 39, 11) and abs(float(policy.sum()) - 1.0) < 1e-4
    move, tree, save = trl.MCTS(cfg, g, net)
    assert np.array_equal(before, g._pack())                      # MCTS works on a copy (ai.py:304)
    root = tree.get_node("root")
    kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
    assert save is True and root.data.visit_count == 24 and sum(k.visit_count for k in kids) == 23
    mask = trl.get_move_matrix(g.players[g.turn], "convolutional")
    assert mask[move[0], move[2], move[1] + 2]
    assert move == max(kids, key=lambda k: (k.visit_count, -kids.index(k))).move      # argmax N, first maximum
    pi = trl.search_statistics(tree)
    assert abs(sum(sum(sum(r) for r in pl) for pl in pi) - 1.0) < 0.01
    assert len(trl.get_move_list(mask, policy)) == int(mask.sum())
