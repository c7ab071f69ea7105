#!/usr/bin/env python3
"""Golden fixtures: whole reference play_game games (ai.py:1570-1699) at BASELINE's MAX_ITER=160 with training=True
(playout cap 400 / 80, root noise, temperature pick) -- the C1 / C3 game loop -- plus the two Config switches round 1
rejected: use_tanh (value range -1..1, ai.py:198-216) and use_random_starting_moves (ai.py:1588-1608), and single
searches with use_tanh.  Same harness as oracle/gen_golden_mcts.py (stub evaluator, engine RNG injected); additionally
    numpy.random.exponential -> engine draw (purpose 6, counter 0):  -scale * log(u)
and for use_tanh the stub's value is 2 v - 1.

use_random_starting_moves with training=True cannot be recorded: under Python >= 3.9 the reference raises
ValueError('Total of weights must be finite') at its first random move -- the 1-simulation search's temperature pick
divides 0 by 0 (ai.py:601-604) and random.choices rejects the NaN weights.  The fixture uses training=False.

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_selfplay160.py  -> tests/golden/selfplay160.npz
"""
import hashlib
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import gen_golden_mcts as gm  # noqa: E402

OUT = gm.OUT
SEED = gm.SEED

GAMES = [  # tag, ruleset, stream, max_iter, training, extra Config kwargs
    ("s2_train160", "s2", 910, 160, True, {}),
    ("s1_train160", "s1", 911, 160, True, {}),
    ("s2_randstart", "s2", 912, 48, False, dict(use_random_starting_moves=True)),
    ("s2_tanh", "s2", 913, 48, True, dict(use_tanh=True)),
    ("s1_randstart_nosoftmax", "s1", 914, 32, False, dict(use_random_starting_moves=True, use_root_softmax=False)),
]


class Tape(gm.SearchTape):
    def exponential(self, scale=1.0):
        return -scale * math.log(gm.unif(self.rng.seed, self.rng.stream, 6, 0))


def main():
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSameConfig
    import gen_golden as gg

    state = {"rng": None, "tanh": False}

    def stub_evaluate(config, game, network):
        v, p = gm.stub_eval_state(rh.pack_game(game, state["rng"], 0, game.ruleset))
        if state["tanh"]:
            v = float(np.float32(2.0) * np.float32(v) - np.float32(1.0))
        return v, p

    ai.evaluate = stub_evaluate
    real_gamma, real_exp, real_mcts = np.random.gamma, np.random.exponential, ai.MCTS
    games = []
    for tag, ruleset, stream, max_iter, training, kw in GAMES:
        rng = rh.EngineRNG(SEED, stream)
        rh.install_rng(mods, rng)
        tape = Tape(rng)
        state["rng"], state["tanh"] = rng, bool(kw.get("use_tanh"))
        ai.random = tape
        np.random.gamma, np.random.exponential = tape.gamma, tape.exponential
        log = []

        def logged_mcts(config, game, net, _log=log, _tape=tape, _rng=rng):
            _tape.ply = sum(p.stats.pieces for p in game.players)
            pre = rh.pack_game(game, _rng, _tape.ply, game.ruleset)
            move, tree, save = real_mcts(config, game, net)
            root = tree.get_node("root")
            kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
            _log.append(dict(ply=_tape.ply, state=pre, move=rh.encode_move(*move), save=int(save), max_iter=config.MAX_ITER,
                             moves=[rh.encode_move(*k.move) for k in kids], visits=[k.visit_count for k in kids]))
            return move, tree, save

        ai.MCTS = logged_mcts
        real_pick = ai.pick_random_move_by_policy
        picks = []

        def logged_pick(tree, _picks=picks):
            mv = real_pick(tree)
            _picks.append(rh.encode_move(*mv))
            return mv

        ai.pick_random_move_by_policy = logged_pick
        try:
            cfg = ai.Config(visual=False, model="pytorch", model_config=AlphaSameConfig(blocks=10, filters=16), MAX_ITER=max_iter,
                            training=training, ruleset=ruleset, save_all=not training, **kw)
            data, stats = ai.play_game(cfg, None)
        finally:
            ai.MCTS, ai.pick_random_move_by_policy = real_mcts, real_pick
            np.random.gamma, np.random.exponential = real_gamma, real_exp
        # the random starting moves: searches with MAX_ITER = 1 whose returned move is replaced by the policy pick
        n_rand = len(picks)
        for k in range(n_rand):
            assert log[k]["max_iter"] == 1
            log[k]["move"], log[k]["save"], log[k]["moves"], log[k]["visits"] = picks[k], 0, [], []
        rec_hash = [hashlib.sha256(json.dumps(r, separators=(",", ":")).encode()).hexdigest() for r in data]
        games.append(dict(tag=tag, ruleset=ruleset, stream=stream, training=training, max_iter=max_iter, log=log, rec_hash=rec_hash,
                          z=[float(r[11]) for r in data], n_rand=n_rand, n_records=len(data),
                          use_tanh=int(bool(kw.get("use_tanh"))), random_start=int(bool(kw.get("use_random_starting_moves"))),
                          root_softmax=int(kw.get("use_root_softmax", True))))
        print("game %-24s: %d plies (%d random starting moves), %d records, z values %s" %
              (tag, len(log), n_rand, len(data), sorted(set(games[-1]["z"]))), flush=True)

    # ---- single searches with use_tanh (eval and training), 160 simulations ----------------------------------
    tr = np.load(os.path.join(OUT, "env_traces.npz"))
    meta = tr["meta"]
    cases = []
    state["tanh"] = True
    for vi, training in enumerate((False, True)):
        for gi in (24, 31, 67, 75):
            npl, s_off = int(meta[gi, 3]), int(meta[gi, 4])
            for frac in (0.2, 0.7):
                si = s_off + int(frac * (npl - 2))
                g, rng, ply = gg.game_from_state(bytes(tr["states"][si]))
                tape = Tape(rng); tape.ply = ply
                state["rng"] = rng
                ai.random = tape
                np.random.gamma = tape.gamma
                try:
                    cfg = ai.Config(visual=False, model="pytorch", model_config=AlphaSameConfig(blocks=10, filters=16),
                                    MAX_ITER=160, training=training, use_tanh=True)
                    move, tree, save = ai.MCTS(cfg, g, None)
                finally:
                    np.random.gamma = real_gamma
                root = tree.get_node("root")
                kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
                cases.append(dict(variant=vi, state_index=si, move=rh.encode_move(*move), save=int(save),
                                  moves=[rh.encode_move(*k.move) for k in kids], visits=[k.visit_count for k in kids],
                                  q=[float(k.value_avg) for k in kids], p=[float(k.policy) for k in kids]))
                print("tanh search %s state %d: move %d, root N %d, min Q %.3f" %
                      ("train" if training else "eval", si, cases[-1]["move"], root.data.visit_count, min(cases[-1]["q"])), flush=True)

    np.savez_compressed(
        os.path.join(OUT, "selfplay160.npz"),
        seed=np.uint64(SEED), numpy_version=np.array(np.__version__),
        game_tag=np.array([g["tag"] for g in games]),
        game_ruleset=np.array([1 if g["ruleset"] == "s1" else 2 for g in games], dtype=np.int32),
        game_stream=np.array([g["stream"] for g in games], dtype=np.int64),
        game_training=np.array([int(g["training"]) for g in games], dtype=np.int32),
        game_max_iter=np.array([g["max_iter"] for g in games], dtype=np.int32),
        game_use_tanh=np.array([g["use_tanh"] for g in games], dtype=np.int32),
        game_random_start=np.array([g["random_start"] for g in games], dtype=np.int32),
        game_root_softmax=np.array([g["root_softmax"] for g in games], dtype=np.int32),
        game_n_rand=np.array([g["n_rand"] for g in games], dtype=np.int32),
        game_plies=np.array([len(g["log"]) for g in games], dtype=np.int32),
        game_records=np.array([g["n_records"] for g in games], dtype=np.int32),
        ply_state=np.frombuffer(b"".join(e["state"] for g in games for e in g["log"]), dtype=np.uint8).reshape(-1, 320),
        ply_move=np.array([e["move"] for g in games for e in g["log"]], dtype=np.uint16),
        ply_save=np.array([e["save"] for g in games for e in g["log"]], dtype=np.int32),
        ply_child_counts=np.array([len(e["moves"]) for g in games for e in g["log"]], dtype=np.int32),
        ply_child_moves=np.array([m for g in games for e in g["log"] for m in e["moves"]], dtype=np.uint16),
        ply_child_visits=np.array([v for g in games for e in g["log"] for v in e["visits"]], dtype=np.int32),
        rec_hash=np.array([h for g in games for h in g["rec_hash"]]),
        rec_z=np.array([z for g in games for z in g["z"]]),
        tanh_states=tr["states"][[c["state_index"] for c in cases]],
        tanh_variant=np.array([c["variant"] for c in cases], dtype=np.int32),
        tanh_move=np.array([c["move"] for c in cases], dtype=np.int32),
        tanh_save=np.array([c["save"] for c in cases], dtype=np.int32),
        tanh_child_counts=np.array([len(c["moves"]) for c in cases], dtype=np.int32),
        tanh_child_moves=np.array([m for c in cases for m in c["moves"]], dtype=np.uint16),
        tanh_child_visits=np.array([v for c in cases for v in c["visits"]], dtype=np.int32),
        tanh_child_q=np.array([v for c in cases for v in c["q"]], dtype=np.float64),
        tanh_child_p=np.array([v for c in cases for v in c["p"]], dtype=np.float64))


if __name__ == "__main__":
    main()
