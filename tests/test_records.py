"""Record writer (host side) against the reference's play_game output: the sha256 of every
record's JSON text (13 fields: features, z, pi; x4 mirror augmentation) must match the hashes
oracle/gen_golden_mcts.py stored, including element types and order."""
import hashlib
import json
import os

import numpy as np
import pytest

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


def test_text_writer_is_byte_identical_to_json_dumps(golden_dir):
    """records.record_texts_for_move / finish_game_texts (templates patched in place, no Python lists) produce
    exactly json.dumps of the list path, and hence the reference's records (sha256 fixtures)"""
    d = np.load(os.path.join(golden_dir, "selfplay.npz"))
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    r_off = np.concatenate([[0], np.cumsum(d["game_records"])])
    for gi in range(d["game_ruleset"].shape[0]):
        training = bool(d["game_training"][gi])
        lists, texts = [[], []], [[], []]
        for k in range(p_off[gi], p_off[gi + 1]):
            if d["ply_save"][k] or not training:
                st = d["ply_state"][k]
                mv, vis = d["ply_child_moves"][c_off[k]:c_off[k + 1]], d["ply_child_visits"][c_off[k]:c_off[k + 1]]
                lists[int(st[312])] += records.records_for_move(st, mv, vis, augment=True)
                texts[int(st[312])] += records.record_texts_for_move(st, mv, vis, augment=True)
        for winner in (0, -1):
            a = records.finish_game([[list(r) for r in lists[0]], [list(r) for r in lists[1]]], winner)
            b = records.finish_game_texts(texts, winner)
            assert len(a) == len(b) == d["game_records"][gi]
            for ra, tb in zip(a, b):
                assert json.dumps(ra).encode() == tb
        # and against the reference's hashes (winner as played: z is part of the record)
        want = d["rec_hash"][r_off[gi]:r_off[gi + 1]]
        z = d["rec_z"][r_off[gi]:r_off[gi + 1]]
        winner = -1 if z[0] == 0.5 else (0 if (z[0] == 1) == (len(lists[0]) > 0) else 1)
        got = records.finish_game_texts(texts, winner)
        for i, tb in enumerate(got):
            h = hashlib.sha256(json.dumps(json.loads(tb), separators=(",", ":")).encode()).hexdigest()
            assert h == str(want[i]), "record %d of game %d" % (i, gi)
    # not augmented, and a visit share that rounds to 0.0 (kept as a float zero; dropped by the mirror)
    st = d["ply_state"][0]
    mv = d["ply_child_moves"][c_off[0]:c_off[1]]
    vis = np.ones(len(mv), dtype=np.int64); vis[0] = 60000
    for aug in (False, True):
        a = records.finish_game([records.records_for_move(st, mv, vis, augment=aug), []], 1, 0, 0.5, 1)
        b = records.finish_game_texts([records.record_texts_for_move(st, mv, vis, augment=aug), []], 1, 0, 0.5, 1)
        assert [json.dumps(r).encode() for r in a] == b


def _fixture_sets(golden_dir, augment=True):
    """the fixture's games as (list-form records, packed moves) per game, z from a made-up winner"""
    d = np.load(os.path.join(golden_dir, "selfplay.npz"))
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    out = []
    for gi in range(d["game_ruleset"].shape[0]):
        gd, gp = [[], []], [[], []]
        for k in range(p_off[gi], p_off[gi + 1]):
            st = d["ply_state"][k]
            mv, vis = d["ply_child_moves"][c_off[k]:c_off[k + 1]], d["ply_child_visits"][c_off[k]:c_off[k + 1]]
            gd[int(st[312])] += records.records_for_move(st, mv, vis, augment=augment)
            gp[int(st[312])].append((st, mv, vis))
        winner = gi % 3 - 1
        lists = records.finish_game(gd, winner, 0, 0.5, 1)
        packed = [(s_, m_, v_, 0.5 if winner == -1 else (1 if winner == pl else 0)) for pl in range(2) for s_, m_, v_ in gp[pl]]
        out.append((lists, packed))
    return out


@pytest.mark.parametrize("augment", [True, False])
def test_packed_set_expands_to_the_same_tensors(golden_dir, augment, tmp_path):
    """records.PackedSet (what make_training_set stores for this build's learner) -> the 13 feature arrays, equal
    element for element and dtype for dtype to torch.tensor over the reference-format records"""
    import torch
    sets = _fixture_sets(golden_dir, augment)
    lists = [r for l, _ in sets for r in l]
    ps = records.PackedSet.from_moves([p for _, pk in sets for p in pk], augment)
    ps.save(str(tmp_path / "0.npz"))
    ps = records.PackedSet.load(str(tmp_path / "0.npz"))
    assert len(ps) == len(lists)
    feats = ps.features()
    ref = list(map(list, zip(*lists)))
    for i in range(13):
        a, b = torch.tensor(ref[i]), torch.from_numpy(np.ascontiguousarray(feats[i]))
        if i == 11:
            a, b = a.type(torch.float), b.type(torch.float)
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), i
    sel = np.array([5, 0, len(ps) - 1, 7, 7])
    for x, y in zip(ps.features(sel), feats):
        assert np.array_equal(x, y[sel])


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
