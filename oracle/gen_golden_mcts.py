#!/usr/bin/env python3
"""Golden fixtures for the search and the game driver: runs the reference's MCTS / play_game
(ai.py:299-659, 1570-1699) with a deterministic stub evaluator and the engine's RNG injected.

TEST INFRASTRUCTURE ONLY; build container only (needs /root/reference).

    python oracle/gen_golden_mcts.py        # writes tests/golden/mcts.npz, tests/golden/selfplay.npz

What is patched (attributes only, no reference file is modified):
  ai.evaluate            -> stub_evaluate: value / policy are hashes of the position (the same
                            function exists in oracle/tetris_oracle_mcts.c and as a test kernel)
  ai.random              -> SearchTape: random() = playout-cap coin, choices() = temperature pick
  numpy.random.gamma     -> the engine's counter-based gamma sampler (root noise, ai.py:488)
  piece_queue.random, player.random -> EngineRNG (bags, garbage columns -- also inside the search)
Recorded with numpy %s semantics (float32 weak-scalar promotion in PUCT, see
oracle/tetris_oracle_mcts.c).
"""
import bisect
import hashlib
import itertools
import json
import math
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
SEED = 20260630
M64 = (1 << 64) - 1
GOLD = 0x9E3779B97F4A7C15


def unif(seed, stream, purpose, ctr):
    k = rh.mix64((seed ^ ((stream * 0xD1342543DE82EF95) & M64)) & M64)
    u = rh.mix64(k ^ ((purpose << 56) & M64) ^ ctr)
    return ((u >> 11) + 0.5) * (1.0 / 9007199254740992.0)


def gamma_draw(seed, stream, base_ctr, a):
    d = (a + 1.0) - 1.0 / 3.0
    c = 1.0 / math.sqrt(9.0 * d)
    for attempt in range(64):
        cc = (base_ctr * 64 + attempt) * 4
        u1, u2 = unif(seed, stream, 5, cc), unif(seed, stream, 5, cc + 1)
        x = math.sqrt(-2.0 * math.log(u1)) * math.cos(6.283185307179586 * u2)
        v = 1.0 + c * x
        if v <= 0.0:
            continue
        v = v * v * v
        u = unif(seed, stream, 5, cc + 2)
        if math.log(u) < 0.5 * x * x + d - d * v + d * math.log(v):
            ub = unif(seed, stream, 5, cc + 3)
            return d * v * math.exp(math.log(ub) / a)
    return 0.0


class SearchTape:
    """stands in for the `random` module inside ai.py (coin ai.py:324, choices ai.py:604)"""

    def __init__(self, rng):
        self.rng = rng
        self.ply = 0

    def random(self):
        return unif(self.rng.seed, self.rng.stream, 3, self.ply)

    def choices(self, population, weights=None, k=1):
        # CPython random.choices: accumulate, bisect over [0, n-1)
        assert k == 1 and weights is not None
        cum = list(itertools.accumulate(weights))
        total = cum[-1] + 0.0
        hi = len(population) - 1
        x = unif(self.rng.seed, self.rng.stream, 4, self.ply) * total
        return [population[bisect.bisect(cum, x, 0, hi)]]

    def gamma(self, shape, scale, size):
        assert scale == 1
        return np.array([gamma_draw(self.rng.seed, self.rng.stream, self.ply * 512 + k, shape)
                         for k in range(size)], dtype=np.float64)


# ---- stub evaluator ----------------------------------------------------------------------------
def stub_hash(state_bytes):
    h = 0x243F6A8885A308D3
    for i in range(2):
        pb = state_bytes[144 * i:144 * (i + 1)]
        words = struct.unpack("<15Q", pb[:120])
        for w in words:
            h = rh.mix64(h ^ w)
        f = struct.unpack(rh.PLAYER_FMT, pb)
        b2b, combo, lvl, piece, held, ql, gl, over = f[48:56]
        meta = ((b2b & 0xFFFF) | ((combo & 0xFFFF) << 16) | (lvl << 32) | (piece << 40) | (held << 44)
                | (ql << 48) | (gl << 53) | (over << 60))
        h = rh.mix64(h ^ meta)
    return rh.mix64(h ^ state_bytes[312])


_IDX = (np.arange(1, 27 * 39 * 11 + 1, dtype=np.uint64) * np.uint64(GOLD))


def _mix64_np(z):
    with np.errstate(over="ignore"):
        z = z + np.uint64(GOLD)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def stub_eval_state(state_bytes):
    h = stub_hash(state_bytes)
    value = np.float32(rh.mix64(h) >> 40) * np.float32(1.0 / 16777216.0)
    k = _mix64_np(np.uint64(h) ^ _IDX)
    u = (k >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    pol = np.where((k & np.uint64(15)) == 0, np.float32(0), u * u).astype(np.float32)
    return float(value), pol.reshape(27, 39, 11)


def main():
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSameConfig
    import gen_golden as gg          # reuses game_from_state / new_game (import runs no generation)

    state = {"rng": None, "tape": None}

    def stub_evaluate(config, game, network):
        buf = rh.pack_game(game, state["rng"], 0, game.ruleset)
        return stub_eval_state(buf)

    ai.evaluate = stub_evaluate
    real_gamma = np.random.gamma

    def make_cfg(training, **kw):
        return ai.Config(visual=False, model="pytorch", model_config=AlphaSameConfig(blocks=10, filters=16),
                         MAX_ITER=160, training=training, **kw)

    def run_search(buf, training, **kw):
        g, rng, ply = gg.game_from_state(buf)
        tape = SearchTape(rng)
        tape.ply = ply
        state["rng"], state["tape"] = rng, tape
        ai.random = tape
        np.random.gamma = tape.gamma
        try:
            cfg = make_cfg(training, **kw)
            move, tree, save = ai.MCTS(cfg, g, None)
        finally:
            np.random.gamma = real_gamma
        root = tree.get_node("root")
        kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
        return dict(move=rh.encode_move(*move), save=int(save), n_nodes=len(tree._nodes),
                    root_visit=root.data.visit_count, garb_ctr=rng.garb_ctr,
                    moves=[rh.encode_move(*k.move) for k in kids], visits=[k.visit_count for k in kids],
                    q=[float(k.value_avg) for k in kids], p=[float(k.policy) for k in kids])

    # ---- single searches on sampled mid-game positions ---------------------------------------
    tr = np.load(os.path.join(OUT, "env_traces.npz"))
    meta = tr["meta"]
    picks = []
    for gi in (24, 27, 31, 35, 64, 67, 70, 75):          # greedy games (line clears, garbage), both rulesets
        npl, s_off = int(meta[gi, 3]), int(meta[gi, 4])
        for frac in (0.0, 0.35, 0.8):
            picks.append(s_off + int(frac * (npl - 2)))
    variants = [
        ("eval", dict(training=False)),
        ("train", dict(training=True)),
        ("train_nosoftmax", dict(training=True, use_root_softmax=False)),
        ("eval_absfpu", dict(training=False, FpuStrategy="absolute", FpuValue=0.4)),
        ("train_forced", dict(training=True, use_forced_playouts_and_policy_target_pruning=True)),
    ]
    cases = []
    for vi, (name, kw) in enumerate(variants):
        for pi, si in enumerate(picks):
            if vi in (2, 3) and pi % 4:
                continue
            if vi == 4 and pi % 2:
                continue
            buf = bytes(tr["states"][si])
            kw2 = dict(kw); training = kw2.pop("training")
            r = run_search(buf, training, **kw2)
            r["variant"], r["state_index"] = vi, si
            cases.append(r)
            print("search %-16s state %5d: move %5d nodes %6d root N %d save %d" %
                  (name, si, r["move"], r["n_nodes"], r["root_visit"], r["save"]), flush=True)
    np.savez_compressed(
        os.path.join(OUT, "mcts.npz"),
        seed=np.uint64(SEED), numpy_version=np.array(np.__version__),
        variant_names=np.array([v[0] for v in variants]),
        states=tr["states"][[c["state_index"] for c in cases]],
        variant=np.array([c["variant"] for c in cases], dtype=np.int32),
        move=np.array([c["move"] for c in cases], dtype=np.int32),
        save=np.array([c["save"] for c in cases], dtype=np.int32),
        n_nodes=np.array([c["n_nodes"] for c in cases], dtype=np.int32),
        root_visit=np.array([c["root_visit"] for c in cases], dtype=np.int32),
        garb_ctr=np.array([c["garb_ctr"] for c in cases], dtype=np.int64),
        child_counts=np.array([len(c["moves"]) for c in cases], dtype=np.int32),
        child_moves=np.array([m for c in cases for m in c["moves"]], dtype=np.uint16),
        child_visits=np.array([v for c in cases for v in c["visits"]], dtype=np.int32),
        child_q=np.array([v for c in cases for v in c["q"]], dtype=np.float64),
        child_p=np.array([v for c in cases for v in c["p"]], dtype=np.float64))

    # ---- full self-play games through the reference's play_game ------------------------------
    games = []
    for ruleset, stream, training in (("s2", 900, True), ("s1", 901, True), ("s2", 902, False)):
        rng = rh.EngineRNG(SEED, stream)
        rh.install_rng(mods, rng)
        tape = SearchTape(rng)
        state["rng"], state["tape"] = rng, tape
        ai.random = tape
        np.random.gamma = tape.gamma
        log = []
        real_mcts = ai.MCTS

        def logged_mcts(config, game, net, _log=log, _tape=tape, _rng=rng):
            _tape.ply = sum(p.stats.pieces for p in game.players)
            pre = rh.pack_game(game, _rng, _tape.ply, game.ruleset)
            move, tree, save = real_mcts(config, game, net)
            root = tree.get_node("root")
            kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
            _log.append(dict(ply=_tape.ply, state=pre, move=rh.encode_move(*move), save=int(save),
                             moves=[rh.encode_move(*k.move) for k in kids], visits=[k.visit_count for k in kids]))
            return move, tree, save

        ai.MCTS = logged_mcts
        try:
            cfg = make_cfg(training, ruleset=ruleset, save_all=not training)
            cfg.MAX_ITER = 48          # keeps the fixture small; the single searches above use 160
            data, stats = ai.play_game(cfg, None)
        finally:
            ai.MCTS = real_mcts
            np.random.gamma = real_gamma
        # records: 11 features + z + pi (ai.py:1399-1413, 1675-1695); hashed, not stored
        rec_hash = [hashlib.sha256(json.dumps(r, separators=(",", ":")).encode()).hexdigest() for r in data]
        zs = [float(r[11]) for r in data]
        games.append(dict(ruleset=ruleset, stream=stream, training=training, log=log, rec_hash=rec_hash, z=zs,
                          app=stats["app"], dspp=stats["dspp"], n_records=len(data)))
        print("game %s stream %d: %d plies, %d records, app %.3f" % (ruleset, stream, len(log), len(data), stats["app"]),
              flush=True)
    np.savez_compressed(
        os.path.join(OUT, "selfplay.npz"),
        seed=np.uint64(SEED), numpy_version=np.array(np.__version__), max_iter=np.int32(48),
        game_ruleset=np.array([1 if g["ruleset"] == "s1" else 2 for g in games], dtype=np.int32),
        game_stream=np.array([g["stream"] for g in games], dtype=np.int64),
        game_training=np.array([int(g["training"]) for g in games], dtype=np.int32),
        game_plies=np.array([len(g["log"]) for g in games], dtype=np.int32),
        game_records=np.array([g["n_records"] for g in games], dtype=np.int32),
        game_app=np.array([g["app"] for g in games]), game_dspp=np.array([g["dspp"] for g in games]),
        ply_state=np.frombuffer(b"".join(e["state"] for g in games for e in g["log"]), dtype=np.uint8).reshape(-1, 320),
        ply_move=np.array([e["move"] for g in games for e in g["log"]], dtype=np.uint16),
        ply_save=np.array([e["save"] for g in games for e in g["log"]], dtype=np.int32),
        ply_child_counts=np.array([len(e["moves"]) for g in games for e in g["log"]], dtype=np.int32),
        ply_child_moves=np.array([m for g in games for e in g["log"] for m in e["moves"]], dtype=np.uint16),
        ply_child_visits=np.array([v for g in games for e in g["log"] for v in e["visits"]], dtype=np.int32),
        rec_hash=np.array([h for g in games for h in g["rec_hash"]]),
        rec_z=np.array([z for g in games for z in g["z"]]))


if __name__ == "__main__":
    main()
