#!/usr/bin/env python3
"""Golden fixture: reference searches with Config.move_algorithm = 'brute-force' (ai.py:84, 414 -> get_move_matrix(player,
algo='brute-force'), move_generation.py:204-243) -- the reference's other generator inside MCTS.  It differs from
'convolutional' only in the T-spin sub-plane of some placements, so the positions are T-piece positions of
tests/golden/movegen_t.npz on which the two generators DISAGREE at the root (a different child list, hence different
priors and visits) plus a few where they agree.  Harness as oracle/gen_golden_mcts.py: stub evaluator, engine RNG injected.

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_mcts_bf.py -> tests/golden/mcts_bf.npz
"""
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import gen_golden_mcts as gm  # noqa: E402

OUT = gm.OUT
SEED = gm.SEED


def state_from_rows(rows, piece, held, q0, stream):
    """a canonical two-player state whose player 0 (to move) has the fixture's board / active / held / next piece; the
    rest of the preview queue is a fixed bag, player 1 an empty board with its own pieces"""
    def player(rows40, piece, held, queue):
        q = 0
        for k, pc in enumerate(queue):
            q |= pc << (4 * k)
        return struct.pack(rh.PLAYER_FMT, *[int(r) for r in rows40], 0, 0, 0, 0, q, 0, 0, 0, -1, 0, 0, piece, held, len(queue), 0, 0)
    queue0 = [int(q0)] + [k for k in (0, 1, 2, 3, 4, 5, 6) if k != int(q0)][:5]
    p0 = player(rows, int(piece), int(held), queue0)
    p1 = player([0] * 40, 2, rh.NO_PIECE, [3, 4, 5, 6, 0, 1])
    tail = struct.pack(rh.GAME_TAIL_FMT, stream, 2, 0, 0, 0, 0, 0, 0, 2)
    return p0 + p1 + tail


def main():
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSameConfig
    import gen_golden as gg

    state = {"rng": None}

    def stub_evaluate(config, game, network):
        return gm.stub_eval_state(rh.pack_game(game, state["rng"], 0, game.ruleset))

    ai.evaluate = stub_evaluate
    d = np.load(os.path.join(OUT, "movegen_t.npz"))
    differ = np.nonzero(d["differ"])[0]
    agree = np.nonzero(d["differ"] == 0)[0]
    picks = list(differ[np.linspace(0, len(differ) - 1, 12).astype(int)]) + list(agree[:4])
    cases = []
    for ci, i in enumerate(picks):
        buf = state_from_rows(d["rows"][i], d["piece"][i], d["held"][i], d["q0"][i], 7000 + ci)
        for algo in ("brute-force", "convolutional"):
            game, rng, ply = gg.game_from_state(buf)
            state["rng"] = rng
            tape = gm.SearchTape(rng); tape.ply = ply
            ai.random = tape
            cfg = ai.Config(visual=False, model="pytorch", model_config=AlphaSameConfig(blocks=10, filters=16), MAX_ITER=48,
                            training=False, move_algorithm=algo)
            move, tree, save = ai.MCTS(cfg, game, None)
            root = tree.get_node("root")
            kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
            cases.append(dict(state=buf, algo=0 if algo == "convolutional" else 1, move=rh.encode_move(*move),
                              moves=[rh.encode_move(*k.move) for k in kids], visits=[int(k.visit_count) for k in kids],
                              q=[float(k.value_avg) for k in kids], p=[float(k.policy) for k in kids]))
        a, b = cases[-2], cases[-1]
        print("position %3d: brute-force %d children move %d | convolutional %d children move %d | child lists %s" %
              (i, len(a["moves"]), a["move"], len(b["moves"]), b["move"], "differ" if a["moves"] != b["moves"] else "equal"), flush=True)
    np.savez_compressed(
        os.path.join(OUT, "mcts_bf.npz"), seed=np.uint64(SEED), numpy_version=np.array(np.__version__), max_iter=np.int32(48),
        states=np.frombuffer(b"".join(c["state"] for c in cases), dtype=np.uint8).reshape(-1, 320),
        algo=np.array([c["algo"] for c in cases], dtype=np.int32), move=np.array([c["move"] for c in cases], dtype=np.int32),
        child_counts=np.array([len(c["moves"]) for c in cases], dtype=np.int32),
        child_moves=np.array([m for c in cases for m in c["moves"]], dtype=np.uint16),
        child_visits=np.array([v for c in cases for v in c["visits"]], dtype=np.int32),
        child_q=np.array([v for c in cases for v in c["q"]], dtype=np.float64),
        child_p=np.array([v for c in cases for v in c["p"]], dtype=np.float64))


if __name__ == "__main__":
    main()
