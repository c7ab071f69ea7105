"""The mask-parallel legal-placement search as the device runs it (csrc/trl_device.h: movegen_closure_wave,
resolve_mixed_wave), stated in Python in scripts/bfs_closure_proto.py, against the reference-generated fixtures: the
closure of the start cell reproduces the reference's FIFO walk (move_generation.py:325-488) for every piece but T and for
the T searches without mixed cells, resolve_mixed settles most of the mixed ones from the first two generations of the
walk, and what is left -- and only that -- is handed to the ordered walk.  (The device itself is checked against the same
fixtures in test_hip_parity.py; this test pins the algorithm's statement and the share of searches left to the walk.)"""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model():
    spec = importlib.util.spec_from_file_location("bfs_closure_proto", os.path.join(ROOT, "scripts", "bfs_closure_proto.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_closure_and_mixed_cell_rule_reproduce_the_reference_lists(golden_dir):
    m = _model()
    d = np.load(os.path.join(golden_dir, "movegen_t.npz"))                  # 360 T positions, 240 of them complex
    off = np.concatenate([[0], np.cumsum(d["conv_counts"])])
    left = settled = 0
    for i in range(len(d["rows"])):
        rows = d["rows"][i].astype(int).tolist()
        piece, held, q0 = int(d["piece"][i]), int(d["held"][i]), int(d["q0"][i])
        p2 = held if held != m.P_NONE else q0
        moves, walk = [], False
        for t, need in [(piece, False)] + ([(p2, True)] if p2 != piece else []):
            r = m.closure_search(rows, t, need)
            if r is None:
                continue
            walk |= r[3]
            settled += int(r[6] > 0 and not r[3])
            moves += m.placements_to_moves(r[0], r[1], r[2], t)
        left += int(walk)
        if not walk:
            assert sorted(moves) == d["conv_flat"][off[i]:off[i + 1]].tolist(), i
    assert settled >= 200 and left <= 20, (settled, left)                   # 225 settled, 14 left to the ordered walk

    d = np.load(os.path.join(golden_dir, "env_traces.npz"))                 # every 16th position of the fixture games
    off = np.concatenate([[0], np.cumsum(d["legal_counts"])])
    checked = 0
    for i in range(0, len(d["states"]), 16):
        st = d["states"][i]
        if st[141] or st[144 + 141]:
            continue
        got, info = m.legal_for_state(st)
        if any(x[1] for x in info):
            continue
        assert got == d["legal_flat"][off[i]:off[i + 1]].tolist(), i
        checked += 1
    assert checked > 350


def test_without_the_mixed_cell_rule_a_fifth_of_the_t_searches_need_the_walk(golden_dir, monkeypatch):
    m = _model()
    monkeypatch.setattr(m, "RESOLVE_MIXED", False)
    d = np.load(os.path.join(golden_dir, "movegen_t.npz"))
    mixed = 0
    for i in range(0, len(d["rows"]), 3):
        rows = d["rows"][i].astype(int).tolist()
        r = m.closure_search(rows, m.PT, False)
        mixed += int(r is not None and r[3])
    assert mixed >= 40                                                       # (these fixtures are T-heavy boards)
