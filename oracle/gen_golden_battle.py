#!/usr/bin/env python3
"""Golden fixture for the gating match: the reference's battle_networks in both forms --
  * _battle_networks_async / aplay_battle_game / amcts (ai.py:2072-2114, 766-996): all games, threshold checked afterwards;
  * the sequential loop with early termination (ai.py:1996-2049) --
between two deterministic stub evaluators ("networks" = salts of the position hash, the functions
to_stub_eval_salted / k_stub_eval implement).  Game i runs on engine RNG stream STREAM0 + i (ai.Game is wrapped so that
a fresh EngineRNG is installed when a game is created; the stub never suspends, so asyncio.gather plays the games one
after the other and the streams do not interleave).

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_battle.py  ->  tests/golden/battle.npz
"""
import asyncio
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import gen_golden_mcts as gm  # noqa: E402

OUT, SEED = gm.OUT, gm.SEED
STREAM0, GAMES, MAX_ITER = 3000, 8, 24
SALT1, SALT2 = 0x1111, 0x2222


def stub_eval_salted(state_bytes, salt):
    """oracle/tetris_oracle_mcts.c to_stub_eval_salted: the stub with h <- mix64(h ^ salt)"""
    h = gm.stub_hash(state_bytes)
    if salt:
        h = rh.mix64(h ^ salt)
    value = np.float32(rh.mix64(h) >> 40) * np.float32(1.0 / 16777216.0)
    k = gm._mix64_np(np.uint64(h) ^ gm._IDX)
    u = (k >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    pol = np.where((k & np.uint64(15)) == 0, np.float32(0), u * u).astype(np.float32)
    return float(value), pol.reshape(27, 39, 11)


def main():
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSameConfig

    cur = {"rng": None, "index": 0, "log": None}
    real_game = ai.Game

    def game_factory(ruleset):                          # ai.Game(...) inside the battle loops
        rng = rh.EngineRNG(SEED, STREAM0 + cur["index"])
        rh.install_rng(mods, rng)
        cur["rng"], cur["index"] = rng, cur["index"] + 1
        cur["log"].append(dict(moves=[]))
        return real_game(ruleset)

    def stub_evaluate(config, game, network):           # sequential form: `network` is the salt
        return stub_eval_salted(rh.pack_game(game, cur["rng"], 0, game.ruleset), int(network))

    async def stub_aevaluate(config, game, evaluator):  # async form: the evaluator carries the salt; never suspends
        return stub_eval_salted(rh.pack_game(game, cur["rng"], 0, game.ruleset), evaluator.salt)

    class StubBatcher:
        def __init__(self, model, config, max_batch=128, timeout_ms=2):
            self.salt = int(model)

        async def start(self):
            pass

        async def stop(self):
            pass

    real_mcts, real_amcts = ai.MCTS, ai.amcts

    def logged_mcts(config, game, net):
        r = real_mcts(config, game, net)
        cur["log"][-1]["moves"].append(rh.encode_move(*r[0]))
        return r

    async def logged_amcts(config, game, evaluator):
        r = await real_amcts(config, game, evaluator)
        cur["log"][-1]["moves"].append(rh.encode_move(*r[0]))
        return r

    ai.Game, ai.evaluate, ai.aevaluate, ai.BatchedEvaluator = game_factory, stub_evaluate, stub_aevaluate, StubBatcher
    ai.MCTS, ai.amcts = logged_mcts, logged_amcts
    cfg = ai.Config(visual=False, model="pytorch", model_config=AlphaSameConfig(blocks=10, filters=16), MAX_ITER=MAX_ITER,
                    training=False, ruleset="s2", batched_inference=True)
    out = {}
    try:
        # ---- concurrent form --------------------------------------------------------------------------------
        cur["index"], cur["log"] = 0, []
        wins, accepted = ai.battle_networks(SALT1, cfg, SALT2, cfg, 0.55, "moreorequal", GAMES)
        alog = cur["log"]
        print("async: wins", wins, "accepted", accepted, "plies", [len(g["moves"]) for g in alog])
        # ---- sequential form with early termination (batched_inference off) -----------------------------------
        cfg2 = cfg.copy(); cfg2.batched_inference = False
        seq = []
        for thr, ttype in ((0.55, "moreorequal"), (0.3, "more"), (0.8, "moreorequal"), (None, "more")):
            cur["index"], cur["log"] = 0, []
            w, acc = ai.battle_networks(SALT1, cfg2, SALT2, cfg2, thr, ttype, GAMES)
            seq.append(dict(thr=-1.0 if thr is None else thr, ttype=ttype, wins=[float(x) for x in w],
                            accepted=-1 if acc is None else int(acc), played=len(cur["log"])))
            for a, b in zip(cur["log"], alog):            # the same games, as far as they were played
                assert a["moves"] == b["moves"]
            print("sequential thr", thr, ttype, "-> wins", w, "accepted", acc, "after", len(cur["log"]), "games")
    finally:
        ai.Game, ai.MCTS, ai.amcts = real_game, real_mcts, real_amcts
    # winners per game from the final states are not kept by the reference; recompute from the win vector is ambiguous,
    # so replay each logged game on the reference env to get its winner
    winners = []
    for i, g in enumerate(alog):
        rng = rh.EngineRNG(SEED, STREAM0 + i)
        rh.install_rng(mods, rng)
        game = real_game("s2")
        game.setup()
        for m in g["moves"]:
            game.make_move(((m >> 10) & 31, (m & 15) - 2, (m >> 4) & 63))
        winners.append(game.winner)
    np.savez_compressed(
        os.path.join(OUT, "battle.npz"), seed=np.uint64(SEED), stream0=np.int64(STREAM0), games=np.int32(GAMES),
        max_iter=np.int32(MAX_ITER), salt1=np.uint64(SALT1), salt2=np.uint64(SALT2),
        game_plies=np.array([len(g["moves"]) for g in alog], dtype=np.int32),
        game_moves=np.array([m for g in alog for m in g["moves"]], dtype=np.uint16),
        game_winner=np.array(winners, dtype=np.int32),
        async_wins=np.array([float(x) for x in wins]), async_accepted=np.int32(-1 if accepted is None else int(accepted)),
        async_threshold=np.float64(0.55),
        seq_threshold=np.array([s["thr"] for s in seq]), seq_type=np.array([s["ttype"] for s in seq]),
        seq_wins=np.array([s["wins"] for s in seq]), seq_accepted=np.array([s["accepted"] for s in seq], dtype=np.int32),
        seq_played=np.array([s["played"] for s in seq], dtype=np.int32))
    print("winners", winners)


if __name__ == "__main__":
    main()
