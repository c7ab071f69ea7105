"""Record writer (host side) against the reference's play_game output: the sha256 of every
record's JSON text (13 fields: features, z, pi; x4 mirror augmentation) must match the hashes
oracle/gen_golden_mcts.py stored, including element types and order."""
import hashlib
import json
import os

import numpy as np

import oracle_py as op
from tetris_rl_amd import records


def test_records_match_reference_play_game(golden_dir):
    d = np.load(os.path.join(golden_dir, "selfplay.npz"))
    seed = int(d["seed"])
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    r_off = np.concatenate([[0], np.cumsum(d["game_records"])])
    for gi in range(d["game_ruleset"].shape[0]):
        training = bool(d["game_training"][gi])
        game_data = [[], []]
        for k in range(p_off[gi], p_off[gi + 1]):
            if d["ply_save"][k] or not training:            # save or config.save_all (ai.py:1613)
                st = d["ply_state"][k]
                turn = int(st[312])
                game_data[turn] += records.records_for_move(
                    st, d["ply_child_moves"][c_off[k]:c_off[k + 1]], d["ply_child_visits"][c_off[k]:c_off[k + 1]],
                    augment=True)
        cfg = op.default_mcts_config(max_iter=int(d["max_iter"]), training=int(training))
        winner = op.play_game_stub(int(d["game_ruleset"][gi]), seed, int(d["game_stream"][gi]), cfg)["winner"]
        data = records.finish_game(game_data, winner)
        assert len(data) == d["game_records"][gi]
        want = d["rec_hash"][r_off[gi]:r_off[gi + 1]]
        for i, rec in enumerate(data):
            assert len(rec) == 13
            h = hashlib.sha256(json.dumps(rec, separators=(",", ":")).encode()).hexdigest()
            assert h == str(want[i]), "record %d of game %d" % (i, gi)
            assert float(rec[11]) == d["rec_z"][r_off[gi] + i]


def test_double_reflection_is_identity(golden_dir):
    """the property the reference's own test_reflections checks (tests.py:61-77)"""
    d = np.load(os.path.join(golden_dir, "selfplay.npz"))
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    for k in (0, 7, 23, 41):
        st = d["ply_state"][k]
        x = records.game_to_x(st)
        grid = x[0].tolist()
        assert records.reflect_grid(records.reflect_grid(grid)) == grid
        pieces = x[1]
        assert np.array_equal(records.reflect_pieces(records.reflect_pieces(pieces)), pieces)
        pi = records.search_statistics(d["ply_child_moves"][c_off[k]:c_off[k + 1]], d["ply_child_visits"][c_off[k]:c_off[k + 1]])
        assert records.reflect_policy(records.reflect_policy(pi)) == pi
