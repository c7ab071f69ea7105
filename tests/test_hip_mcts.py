"""GPU search + game driver (through the C ABI) against the reference's MCTS / play_game fixtures
and the oracle, using the deterministic stub evaluator on the device.  Visit counts, chosen moves,
save flags and per-ply states are compared exactly; Q / P to 1e-12 (device libm vs glibc in
log/exp of the root softmax and the gamma sampler)."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_py as op
import tetris_rl_amd as trl
from tetris_rl_amd import records

pytestmark = pytest.mark.gpu

VARIANTS = {
    "eval": dict(training=0),
    "train": dict(training=1),
    "train_nosoftmax": dict(training=1, use_root_softmax=0),
    "eval_absfpu": dict(training=0, fpu_reduction=0, fpu_value=0.4),
    "train_forced": dict(training=1, use_forced=1),
}


class StubSearch:
    """ticks an Engine with the on-device stub evaluator"""

    def __init__(self, eng, cfg):
        import torch
        self.torch = torch
        self.eng = eng
        n = eng.n_games
        dev = torch.device("cuda", 0)
        self.grids = torch.zeros((2, n, 400), dtype=torch.float32, device=dev)
        self.vec = torch.zeros((n, 105), dtype=torch.float32, device=dev)
        self.values = torch.zeros(n, dtype=torch.float32, device=dev)
        self.policy = torch.zeros((n, 11583), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        eng.mcts_create(cfg)
        L = trl.load_library()
        L.trl_debug_stub_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.L = L

    def tick(self):
        e = self.eng
        e.mcts_select(self.grids.data_ptr(), self.vec.data_ptr())
        assert self.L.trl_debug_stub_eval(e._h, self.values.data_ptr(), self.policy.data_ptr()) == 0
        e.mcts_expand(self.values.data_ptr(), self.policy.data_ptr())


def test_single_searches_match_reference(golden_dir):
    d = np.load(os.path.join(golden_dir, "mcts.npz"))
    names = [str(x) for x in d["variant_names"]]
    off = np.concatenate([[0], np.cumsum(d["child_counts"])])
    for vi, name in enumerate(names):
        idx = np.nonzero(d["variant"] == vi)[0]
        eng = trl.Engine(len(idx), seed=int(d["seed"]))
        eng.set_state(d["states"][idx])
        s = StubSearch(eng, trl.mcts_config(max_iter=160, auto_reset=0, **VARIANTS[name]))
        first = {}
        for t in range(400):
            s.tick()
            recs = eng.mcts_drain_records()
            for r in recs:
                if r["kind"] == trl.REC_MOVE and int(r["slot"]) not in first:
                    first[int(r["slot"])] = r.copy()
            if len(first) == len(idx):
                break
        assert len(first) == len(idx)
        for j, ci in enumerate(idx):
            r = first[j]
            sl = slice(off[ci], off[ci + 1])
            n = int(r["n_children"])
            tag = "%s case %d" % (name, ci)
            assert n == d["child_counts"][ci], tag
            assert np.array_equal(r["moves"][:n], d["child_moves"][sl]), tag
            assert np.array_equal(r["visits"][:n].astype(np.int32), d["child_visits"][sl]), tag
            assert int(r["chosen"]) == d["move"][ci] and int(r["saved"]) == d["save"][ci], tag
            assert np.array_equal(r["state"], d["states"][ci]), tag
        st = eng.mcts_stats()
        assert st["overflow"] == 0 and st["records_dropped"] == 0


def test_root_q_and_p_after_full_search(golden_dir):
    """one eval-mode search stopped right after its last simulation: Q and P of the root children"""
    d = np.load(os.path.join(golden_dir, "mcts.npz"))
    off = np.concatenate([[0], np.cumsum(d["child_counts"])])
    idx = np.nonzero(d["variant"] == 0)[0][:6]
    eng = trl.Engine(len(idx), seed=int(d["seed"]))
    eng.set_state(d["states"][idx])
    # max_iter 161: after 160 ticks the trees equal the reference's 160-simulation trees
    s = StubSearch(eng, trl.mcts_config(max_iter=161, training=0, auto_reset=0))
    for t in range(160):
        s.tick()
    counts, moves, visits, q, p = eng.mcts_root_children()
    for j, ci in enumerate(idx):
        sl = slice(off[ci], off[ci + 1])
        n = counts[j]
        assert np.array_equal(visits[j, :n], d["child_visits"][sl])
        assert np.allclose(q[j, :n], d["child_q"][sl], rtol=1e-12, atol=1e-15)
        assert np.allclose(p[j, :n], d["child_p"][sl], rtol=1e-12, atol=1e-300)


def _play_until_done(eng, s, max_ticks):
    out = []
    for t in range(max_ticks):
        s.tick()
        if t % 64 == 63 or t == max_ticks - 1:
            out.append(eng.mcts_drain_records())
            if any((r["kind"] == trl.REC_GAME_END).any() for r in out if len(r)):
                ends = sum(int((r["kind"] == trl.REC_GAME_END).sum()) for r in out)
                if ends == eng.n_games:
                    break
    return np.concatenate(out) if out else np.zeros(0, dtype=trl.RECORD_DTYPE)


def test_selfplay_games_match_reference_play_game(golden_dir):
    """three whole reference games (play_game + MCTS, stub evaluator): every ply's searched
    position, chosen move, save flag, root (move, N) vector -- and the 13-field records rebuilt
    from the device's record stream hash to the reference's JSON."""
    d = np.load(os.path.join(golden_dir, "selfplay.npz"))
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    r_off = np.concatenate([[0], np.cumsum(d["game_records"])])
    for gi in range(d["game_ruleset"].shape[0]):
        training = int(d["game_training"][gi])
        eng = trl.Engine(1, ruleset="s1" if d["game_ruleset"][gi] == 1 else "s2", seed=int(d["seed"]))
        eng.reset(np.array([d["game_stream"][gi]], dtype=np.uint32))
        s = StubSearch(eng, trl.mcts_config(max_iter=int(d["max_iter"]), training=training, save_all=int(not training),
                                            auto_reset=0))
        recs = _play_until_done(eng, s, 200 * 130)
        mv = recs[recs["kind"] == trl.REC_MOVE]
        mv = mv[np.argsort(mv["ply"], kind="stable")]
        end = recs[recs["kind"] == trl.REC_GAME_END]
        assert len(end) == 1 and len(mv) == d["game_plies"][gi]
        game_data = [[], []]
        for k in range(len(mv)):
            fk = p_off[gi] + k
            a, b = mv[k]["state"].copy(), d["ply_state"][fk].copy()
            a[304:308] = 0; b[304:308] = 0
            assert np.array_equal(a, b), "searched position, game %d ply %d" % (gi, k)
            n = int(mv[k]["n_children"])
            assert int(mv[k]["chosen"]) == d["ply_move"][fk]
            assert int(mv[k]["saved"]) == (d["ply_save"][fk] or (not training))
            assert np.array_equal(mv[k]["moves"][:n], d["ply_child_moves"][c_off[fk]:c_off[fk + 1]])
            assert np.array_equal(mv[k]["visits"][:n].astype(np.int32), d["ply_child_visits"][c_off[fk]:c_off[fk + 1]])
            if mv[k]["saved"]:
                game_data[int(mv[k]["state"][312])] += records.records_for_move(mv[k]["state"], mv[k]["moves"][:n], mv[k]["visits"][:n])
        data = records.finish_game(game_data, int(end[0]["winner"]))
        want = d["rec_hash"][r_off[gi]:r_off[gi + 1]]
        assert len(data) == len(want)
        for i, rec in enumerate(data):
            assert hashlib.sha256(json.dumps(rec, separators=(",", ":")).encode()).hexdigest() == str(want[i])


def test_many_slots_match_oracle():
    """96 concurrent games (mixed search lengths from playout-cap randomisation) played to the end on
    the device == the oracle's play_game, game by game: plies, moves, winner."""
    n, seed = 96, 4242
    cfgk = dict(max_iter=24, training=1)
    eng = trl.Engine(n, ruleset="s2", seed=seed)
    eng.reset()
    s = StubSearch(eng, trl.mcts_config(auto_reset=0, **cfgk))
    recs = _play_until_done(eng, s, 60 * 250)
    assert eng.mcts_stats()["overflow"] == 0
    for g in range(0, n, 5):
        o = op.play_game_stub(2, seed, g, op.default_mcts_config(**cfgk))
        mine = recs[(recs["slot"] == g) & (recs["kind"] == trl.REC_MOVE)]
        mine = mine[np.argsort(mine["ply"], kind="stable")]
        end = recs[(recs["slot"] == g) & (recs["kind"] == trl.REC_GAME_END)]
        assert len(mine) == o["plies"] and len(end) == 1
        assert np.array_equal(mine["chosen"], o["ply_move"])
        assert int(end[0]["winner"]) == o["winner"]
        assert np.array_equal(end[0]["state"][:288], o["final"][:288])


def _run_ticks(n, seed, cfg, ticks, streams=None):
    eng = trl.Engine(n, ruleset="s2", seed=seed)
    eng.reset(streams)
    s = StubSearch(eng, cfg)
    out = []
    for t in range(ticks):
        s.tick()
        if t % 128 == 127 or t == ticks - 1:
            out.append(eng.mcts_drain_records())
    st = eng.mcts_stats()
    eng.close()
    r = np.concatenate(out)
    return r[np.lexsort((r["kind"], r["ply"], r["slot"]))], st


def test_c3_full_size_properties():
    """BASELINE config C3 at full size -- 1024 concurrent games, MAX_ITER=160, training=True (playout cap 400/80,
    root noise, temperature) -- with the deterministic stub evaluator, through size-independent properties:
    determinism, slot independence (a 256-slot engine on streams 512..767 replays those slots exactly), the
    per-record invariants of a search, and the oracle's play_game on a sample of the games."""
    n, seed, ticks = 1024, 777, 1500
    cfgk = dict(max_iter=160, training=1)
    a, st = _run_ticks(n, seed, trl.mcts_config(auto_reset=0, **cfgk), ticks)
    assert st["overflow"] == 0 and st["records_dropped"] == 0 and st["simulations"] == n * ticks
    moves = a[a["kind"] == trl.REC_MOVE]
    assert len(moves) > 6 * n                                   # several moves per game were searched
    b, _ = _run_ticks(n, seed, trl.mcts_config(auto_reset=0, **cfgk), ticks)
    assert a.tobytes() == b.tobytes()                           # deterministic
    c, _ = _run_ticks(256, seed, trl.mcts_config(auto_reset=0, **cfgk), ticks, streams=np.arange(512, 768, dtype=np.uint32))
    sub = a[(a["slot"] >= 512) & (a["slot"] < 768)].copy()
    sub["slot"] -= 512
    assert sub.tobytes() == c.tobytes()                         # a slot's games depend on (seed, stream) only
    # search invariants (ai.py:299-659): visits of the root children sum to simulations - 1 of a 400- or 80-
    # simulation search; the chosen move is a child; plies count up by one per slot
    nch = moves["n_children"].astype(np.int64)
    vis = np.array([int(m["visits"][:k].sum()) for m, k in zip(moves, nch)])
    assert set(np.unique(vis)) <= {399, 79}
    assert all(m["chosen"] in m["moves"][:k] for m, k in zip(moves[::37], nch[::37]))
    assert (moves["saved"] == (vis == 399)).all()               # only the full searches are saved (ai.py:1613)
    for g in range(0, n, 128):
        mine = moves[moves["slot"] == g]
        assert np.array_equal(mine["ply"], np.arange(len(mine)))
        o = op.play_game_stub(2, seed, g, op.default_mcts_config(**cfgk))
        k = min(len(mine), o["plies"])
        assert k >= 5 and np.array_equal(mine["chosen"][:k], o["ply_move"][:k])
        assert np.array_equal(mine["state"][:k], o["ply_state"][:k])


def test_c5_search_shape_matches_oracle():
    """BASELINE config C5's search shape -- MAX_ITER=800 (playout cap 2000/400), forced playouts + policy-target
    pruning -- on 128 slots with the stub evaluator: the first moves of sampled games equal the oracle's, the
    pruned visit vectors included."""
    n, seed, ticks = 128, 31337, 3000
    cfgk = dict(max_iter=800, training=1, use_forced=1)
    a, st = _run_ticks(n, seed, trl.mcts_config(auto_reset=0, **cfgk), ticks)
    assert st["overflow"] == 0 and st["records_dropped"] == 0
    moves = a[a["kind"] == trl.REC_MOVE]
    for g in (0, 77):
        mine = moves[moves["slot"] == g]
        o = op.play_game_stub(2, seed, g, op.default_mcts_config(**cfgk))
        k = min(len(mine), o["plies"])
        assert k >= 2 and np.array_equal(mine["chosen"][:k], o["ply_move"][:k])
        off = np.concatenate([[0], np.cumsum(o["ply_nchild"])])
        for i in range(k):
            nc = int(mine["n_children"][i])
            assert nc == o["ply_nchild"][i]
            assert np.array_equal(mine["moves"][i][:nc], o["child_moves"][off[i]:off[i + 1]])
            assert np.array_equal(mine["visits"][i][:nc], o["child_visits"][off[i]:off[i + 1]])


def test_oversized_tree_is_refused_cleanly():
    """a tree that cannot fit in HBM: trl_mcts_create fails with an error (no crash, no leak of the partial
    allocations) and the engine takes a normal search afterwards"""
    eng = trl.Engine(64, ruleset="s2", seed=1)
    eng.reset()
    with pytest.raises(trl.EngineError):
        eng.mcts_create(trl.mcts_config(max_iter=16, max_nodes=2_000_000_000))
    s = StubSearch(eng, trl.mcts_config(max_iter=16, training=0, auto_reset=0))
    for _ in range(40):
        s.tick()
    assert eng.mcts_stats()["simulations"] == 40 * 64


def test_record_buffer_overflow_is_counted_not_silent():
    """more finished searches than the device record buffer holds between drains: the surplus is counted in
    stats.records_dropped (the drivers turn that into an error), nothing is overwritten"""
    eng = trl.Engine(64, ruleset="s2", seed=3)
    eng.reset()
    s = StubSearch(eng, trl.mcts_config(max_iter=8, training=0, auto_reset=0, record_capacity=16))
    for _ in range(8):
        s.tick()
    st = eng.mcts_stats()
    recs = eng.mcts_drain_records()
    assert st["moves"] == 64 and len(recs) == 16 and st["records_dropped"] == 64 - 16
    assert len(set(recs["slot"].tolist())) == 16


def test_auto_reset_episodes_match_oracle_games():
    """auto_reset=1 (more games than slots, as make_training_set plays them): when a game ends its slot is set up
    again on RNG stream + 2^24; every (slot, episode) game equals the oracle's play_game on that stream"""
    n, seed = 6, 909
    cfgk = dict(max_iter=16, training=1)
    eng = trl.Engine(n, ruleset="s2", seed=seed)
    eng.reset()
    s = StubSearch(eng, trl.mcts_config(auto_reset=1, **cfgk))
    chunks, ends = [], 0
    for t in range(200000):
        s.tick()
        if t % 256 == 255:
            r = eng.mcts_drain_records()
            chunks.append(r)
            ends += int((r["kind"] == trl.REC_GAME_END).sum())
            if ends >= 3 * n + 6:
                break
    recs = np.concatenate(chunks)
    assert eng.mcts_stats()["overflow"] == 0
    checked = 0
    for slot in range(n):
        for ep in range(3):
            mine = recs[(recs["slot"] == slot) & (recs["episode"] == ep)]
            end = mine[mine["kind"] == trl.REC_GAME_END]
            if len(end) == 0:
                continue
            mv = mine[mine["kind"] == trl.REC_MOVE]
            mv = mv[np.argsort(mv["ply"], kind="stable")]
            o = op.play_game_stub(2, seed, slot + (ep << 24), op.default_mcts_config(**cfgk))
            assert len(mv) == o["plies"] and np.array_equal(mv["chosen"], o["ply_move"]), (slot, ep)
            assert int(end[0]["winner"]) == o["winner"]
            checked += 1
    assert checked >= 2 * n


def test_games_at_max_iter_160_tanh_and_random_start_match_reference(golden_dir):
    """Whole reference play_game games replayed on the device (tests/golden/selfplay160.npz): two at BASELINE's
    MAX_ITER=160 with training=True (playout cap 400 / 80 over a whole game, root noise, temperature pick: the C1 / C3
    game loop), one with Config.use_tanh (value range -1..1), two with Config.use_random_starting_moves (one of them
    without the root softmax: float32-typed priors in pick_random_move_by_policy).  Every ply: searched position, move,
    save flag, root (move, N) vector; and the 13-field records hash to the reference's JSON."""
    d = np.load(os.path.join(golden_dir, "selfplay160.npz"))
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    r_off = np.concatenate([[0], np.cumsum(d["game_records"])])
    for gi in range(d["game_ruleset"].shape[0]):
        training, tanh = int(d["game_training"][gi]), int(d["game_use_tanh"][gi])
        tag = str(d["game_tag"][gi])
        eng = trl.Engine(1, ruleset="s1" if d["game_ruleset"][gi] == 1 else "s2", seed=int(d["seed"]))
        eng.reset(np.array([d["game_stream"][gi]], dtype=np.uint32))
        s = StubSearch(eng, trl.mcts_config(max_iter=int(d["game_max_iter"][gi]), training=training, save_all=int(not training),
                                            auto_reset=0, use_tanh=tanh, random_start=int(d["game_random_start"][gi]),
                                            use_root_softmax=int(d["game_root_softmax"][gi])))
        recs = _play_until_done(eng, s, 420 * 130)
        mv = recs[recs["kind"] == trl.REC_MOVE]
        mv = mv[np.argsort(mv["ply"], kind="stable")]
        end = recs[recs["kind"] == trl.REC_GAME_END]
        assert len(end) == 1 and len(mv) == d["game_plies"][gi], tag
        assert int(mv["random_move"].sum()) == d["game_n_rand"][gi] and mv["random_move"][:int(d["game_n_rand"][gi])].all(), tag
        game_data = [[], []]
        for k in range(len(mv)):
            fk = p_off[gi] + k
            a, b = mv[k]["state"].copy(), d["ply_state"][fk].copy()
            a[304:308] = 0; b[304:308] = 0
            assert np.array_equal(a, b), "searched position, %s ply %d" % (tag, k)
            assert int(mv[k]["chosen"]) == d["ply_move"][fk], "%s ply %d" % (tag, k)
            if mv[k]["random_move"]:
                assert not mv[k]["saved"] and not mv[k]["visits"].any()
                continue
            n = int(mv[k]["n_children"])
            assert int(mv[k]["saved"]) == (d["ply_save"][fk] or (not training))
            assert np.array_equal(mv[k]["moves"][:n], d["ply_child_moves"][c_off[fk]:c_off[fk + 1]])
            assert np.array_equal(mv[k]["visits"][:n].astype(np.int32), d["ply_child_visits"][c_off[fk]:c_off[fk + 1]])
            if mv[k]["saved"]:
                game_data[int(mv[k]["state"][312])] += records.records_for_move(mv[k]["state"], mv[k]["moves"][:n], mv[k]["visits"][:n])
        vmin, vmid, vmax = (-1, 0, 1) if tanh else (0, 0.5, 1)
        data = records.finish_game(game_data, int(end[0]["winner"]), vmin, vmid, vmax)
        want = d["rec_hash"][r_off[gi]:r_off[gi + 1]]
        assert len(data) == len(want), tag
        for i, rec in enumerate(data):
            assert hashlib.sha256(json.dumps(rec, separators=(",", ":")).encode()).hexdigest() == str(want[i]), tag


def test_tanh_single_searches_match_reference(golden_dir):
    """160-simulation reference searches with Config.use_tanh (ai.py:198-216: value_min -1, negate_value -v): root (move, N)
    vectors, chosen move and save flag exact; then Q / P of the root children to 1e-12 on the eval-mode cases"""
    d = np.load(os.path.join(golden_dir, "selfplay160.npz"))
    off = np.concatenate([[0], np.cumsum(d["tanh_child_counts"])])
    for vi in (0, 1):
        idx = np.nonzero(d["tanh_variant"] == vi)[0]
        eng = trl.Engine(len(idx), seed=int(d["seed"]))
        eng.set_state(d["tanh_states"][idx])
        s = StubSearch(eng, trl.mcts_config(max_iter=160, training=vi, auto_reset=0, use_tanh=1))
        first = {}
        for t in range(400):
            s.tick()
            for r in eng.mcts_drain_records():
                if r["kind"] == trl.REC_MOVE and int(r["slot"]) not in first:
                    first[int(r["slot"])] = r.copy()
            if len(first) == len(idx):
                break
        assert len(first) == len(idx)
        for j, ci in enumerate(idx):
            r, sl = first[j], slice(off[ci], off[ci + 1])
            n = int(r["n_children"])
            assert np.array_equal(r["moves"][:n], d["tanh_child_moves"][sl])
            assert np.array_equal(r["visits"][:n].astype(np.int32), d["tanh_child_visits"][sl])
            assert int(r["chosen"]) == d["tanh_move"][ci] and int(r["saved"]) == d["tanh_save"][ci]
    idx = np.nonzero(d["tanh_variant"] == 0)[0]
    eng = trl.Engine(len(idx), seed=int(d["seed"]))
    eng.set_state(d["tanh_states"][idx])
    s = StubSearch(eng, trl.mcts_config(max_iter=161, training=0, auto_reset=0, use_tanh=1))
    for t in range(160):
        s.tick()
    counts, moves, visits, q, p = eng.mcts_root_children()
    for j, ci in enumerate(idx):
        sl = slice(off[ci], off[ci + 1])
        assert np.array_equal(visits[j, :counts[j]], d["tanh_child_visits"][sl])
        assert np.allclose(q[j, :counts[j]], d["tanh_child_q"][sl], rtol=1e-12, atol=1e-15) and d["tanh_child_q"][sl].min() < 0
        assert np.allclose(p[j, :counts[j]], d["tanh_child_p"][sl], rtol=1e-12, atol=1e-300)


def test_search_limits_are_refused():
    """a search too long for the record's uint16 visit counts, or more than 256 episodes per slot, is refused up front"""
    eng = trl.Engine(4, ruleset="s2", seed=1)
    eng.reset()
    with pytest.raises(trl.EngineError):
        eng.mcts_create(trl.mcts_config(max_iter=40000, training=1))           # long searches: 5 * 40000 / 2 > 65535
    with pytest.raises(trl.EngineError):
        eng.mcts_create(trl.mcts_config(max_iter=8, episodes_base=300))
    eng.mcts_create(trl.mcts_config(max_iter=8, episodes_base=2, episodes_extra=1))


def test_a_quota_of_exactly_256_games_ends_cleanly():
    """episode e of a slot runs on RNG stream + e * 2^24 (uint32): 256 episodes per slot are the most a quota may ask for,
    and a slot that finishes its 256th game on that quota stops without tripping the stream-wrap guard (the guard is for
    slots that would go ON); without a quota the same slot is stopped and counted as an overflow"""
    for quota, want_overflow in ((256, 0), (0, 1)):
        eng = trl.Engine(2, ruleset="s2", seed=4242)
        eng.reset()
        st = eng.get_state()
        st[:, 308:312] = np.frombuffer(np.uint32(254).tobytes(), dtype=np.uint8)     # trl_game_state.episode: two games to go
        eng.set_state(st)
        s = StubSearch(eng, trl.mcts_config(max_iter=4, training=0, auto_reset=1, episodes_base=quota, episodes_extra=0))
        ends = []
        for t in range(60000):
            s.tick()
            if t % 128 == 127:
                r = eng.mcts_drain_records()
                ends += [int(e) for e in r["episode"][r["kind"] == trl.REC_GAME_END]]
                if len(ends) >= 4:
                    break
        for _ in range(8):
            s.tick()                                   # idle slots stay idle
        stats = eng.mcts_stats()
        assert sorted(ends) == [254, 254, 255, 255], ends
        assert stats["games"] == 4 and bool(stats["overflow"]) == bool(want_overflow), (quota, stats)
        eng.close()


def test_search_with_the_brute_force_generator_matches_reference(golden_dir):
    """Config.move_algorithm = 'brute-force' inside the search (ai.py:84, 414; k_select_bf): root children, visits and
    chosen moves of the reference's searches on T-piece positions where its two generators disagree -- and the same
    positions with 'convolutional'; also through the package's Config / SelfPlay path (three-kernel tick)"""
    d = np.load(os.path.join(golden_dir, "mcts_bf.npz"))
    off = np.concatenate([[0], np.cumsum(d["child_counts"])])
    for algo in (1, 0):
        idx = np.nonzero(d["algo"] == algo)[0]
        eng = trl.Engine(len(idx), seed=int(d["seed"]))
        eng.set_state(d["states"][idx])
        s = StubSearch(eng, trl.mcts_config(max_iter=int(d["max_iter"]), training=0, auto_reset=0, move_algorithm=algo))
        for t in range(int(d["max_iter"])):
            s.tick()
        recs = eng.mcts_drain_records()
        recs = recs[recs["kind"] == trl.REC_MOVE]
        recs = recs[np.argsort(recs["slot"], kind="stable")]
        assert len(recs) == len(idx) and eng.mcts_stats()["overflow"] == 0
        for j, ci in enumerate(idx):
            sl = slice(off[ci], off[ci + 1])
            n = int(recs["n_children"][j])
            assert n == d["child_counts"][ci] and np.array_equal(recs["moves"][j][:n], d["child_moves"][sl]), (algo, ci)
            assert np.array_equal(recs["visits"][j][:n].astype(np.int32), d["child_visits"][sl]), (algo, ci)
            assert int(recs["chosen"][j]) == d["move"][ci], (algo, ci)
    with pytest.raises(trl.EngineError):
        trl.Engine(1, seed=1).mcts_create(trl.mcts_config(move_algorithm=2))
    # Config(move_algorithm='brute-force') -> SelfPlay: plays, on the three-kernel tick
    import torch
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=8, training=True, move_algorithm="brute-force")
    sp = trl.SelfPlay(cfg, trl.AlphaSame(mc), 16, seed=3)
    assert not sp.fused_select and sp.native_graph
    sp.tick(64)
    st = sp.stats()
    assert st["simulations"] == 64 * 16 and st["overflow"] == 0 and st["moves"] > 16
