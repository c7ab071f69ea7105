#!/usr/bin/env python3
"""Golden fixture, round 3: the reference's nets on 64 positions chosen for what the split-f16 towers are sensitive to
(activation magnitude: tall stacks, garbage rows, nearly full and nearly empty boards, late-game piles of both players),
and whole reference searches with the REAL 10-block / 16-filter net (ai.py:299-659 MCTS with evaluate_pytorch,
ai.py:1296-1319) at MAX_ITER = 160 on 32 of them.

tests/golden/net.npz pins every one of the 11 583 logits on 8 early / mid-game positions; here, for 64 positions and four
nets, the value, 768 sampled logits (fixed index set), and the sum and L2 norm of all logits are stored -- the policy head
is a dense layer over the same 521-vector, so any error in the tower shows in every logit alike.

    b10f16      weights of tests/golden/net.npz (BASELINE C3's architecture)
    b10f16tanh  formula weights, use_tanh            (net_fixture.formula_state_dict, salt as in gen_golden_net_deep.py)
    b20f64      formula weights: BASELINE C5's net at full depth
    b2f32       weights of tests/golden/net_wide.npz

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_net64.py  -> tests/golden/net64.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import net_fixture  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
N_SAMPLED = 768
N_SEARCH = 32


def sampled_indices():
    """768 distinct policy indices from the engine's hash (the same on both sides; stored in the fixture as well)"""
    idx, k = [], 0
    seen = set()
    while len(idx) < N_SAMPLED:
        v = int(net_fixture._mix64(np.array([0xABCD0000 + k], dtype=np.uint64))[0] % np.uint64(11583))
        k += 1
        if v not in seen:
            seen.add(v)
            idx.append(v)
    return np.array(sorted(idx), dtype=np.int64)


def heights(state):
    """(height of player 0's stack, of player 1's, pending garbage of both, filled cells of both)"""
    out = []
    for pl in (0, 144):
        rows = np.frombuffer(bytes(state[pl:pl + 80]), dtype="<u2")
        nz = np.nonzero(rows)[0]
        out.append(40 - int(nz[0]) if len(nz) else 0)
    cells = [int(sum(bin(int(r)).count("1") for r in np.frombuffer(bytes(state[pl:pl + 80]), dtype="<u2"))) for pl in (0, 144)]
    return out[0], out[1], int(state[140]) + int(state[144 + 140]), cells[0] + cells[1]


def pick_positions(tr):
    """64 positions of the greedy-driver games (lines, garbage, tall stacks) and a few uniform ones: the tallest, the
    fullest, the ones with most pending garbage, and an even spread over game phases -- terminal positions excluded"""
    st = tr["states"]
    meta = tr["meta"]
    cand = []
    for gi in range(len(meta)):
        npl, off = int(meta[gi, 3]), int(meta[gi, 4])
        for k in range(0, npl):                              # state k = before ply k; the last state of a game is terminal
            s = st[off + k]
            if s[141] or s[144 + 141]:                       # game_over of either player
                continue
            cand.append((off + k, int(meta[gi, 2])) + heights(s))
    cand = np.array(cand, dtype=np.int64)                    # index, driver, h0, h1, garbage, cells
    chosen = []

    def take(order, n):
        for i in order:
            if len([c for c in chosen if abs(c - int(cand[i, 0])) < 3]) == 0:
                chosen.append(int(cand[i, 0]))
                n -= 1
                if n == 0:
                    return

    take(np.argsort(-(cand[:, 2] + cand[:, 3]), kind="stable"), 16)      # tallest pairs of stacks
    take(np.argsort(-cand[:, 5], kind="stable"), 16)                     # most filled cells (garbage-heavy)
    take(np.argsort(-cand[:, 4], kind="stable"), 8)                      # most pending garbage
    take(np.argsort(cand[:, 5], kind="stable"), 4)                       # emptiest
    greedy = np.nonzero(cand[:, 1] == 1)[0]
    take(greedy[np.linspace(0, len(greedy) - 1, 40).astype(int)], 64 - len(chosen))
    assert len(chosen) == 64
    return np.array(sorted(chosen), dtype=np.int64)


def features(ai, torch, game):
    X = []
    for feature in ai.game_to_X(game):                        # evaluate_pytorch's tensor construction (ai.py:1296-1313)
        t = torch.tensor(feature)
        if t.size() == (40, 10):
            t = t.unsqueeze(dim=0).type(torch.float)
        X.append(t.unsqueeze(dim=0))
    return X


def main():
    import torch
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSame, AlphaSameConfig
    import gen_golden as gg

    torch.set_num_threads(8)
    tr = np.load(os.path.join(OUT, "env_traces.npz"))
    picks = pick_positions(tr)
    states = tr["states"][picks]
    sidx = sampled_indices()
    out = {"states": states, "state_index": picks, "sampled": sidx, "torch_version": np.array(torch.__version__),
           "numpy_version": np.array(np.__version__)}
    hs = np.array([heights(s) for s in states])
    print("positions: stack heights %d..%d / %d..%d, pending garbage up to %d, filled cells %d..%d" %
          (hs[:, 0].min(), hs[:, 0].max(), hs[:, 1].min(), hs[:, 1].max(), hs[:, 2].max(), hs[:, 3].min(), hs[:, 3].max()))

    def fixture_net(tag, blocks, filters, src):
        net = AlphaSame(AlphaSameConfig(blocks=blocks, filters=filters))
        d = np.load(os.path.join(OUT, src))
        w, b = net_fixture.policy_head()
        sd = {}
        for k in net.state_dict().keys():
            sd[k] = torch.from_numpy(w) if k == "policy_head.weight" else torch.from_numpy(b) if k == "policy_head.bias" \
                else torch.from_numpy(d["%s/%s" % (tag, k)])
        net.load_state_dict(sd)
        return net.eval()

    def formula_net(blocks, filters, salt, use_tanh):
        net = AlphaSame(AlphaSameConfig(blocks=blocks, filters=filters), use_tanh=use_tanh)
        new = net_fixture.formula_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, salt=salt)
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in new.items()})
        return net.eval()

    nets = [("b10f16", fixture_net("b10f16", 10, 16, "net.npz")),
            ("b10f16tanh", formula_net(10, 16, 1016, True)),
            ("b20f64", formula_net(20, 64, 2064, False)),
            ("b2f32", fixture_net("b2f32", 2, 32, "net_wide.npz"))]
    games = [gg.game_from_state(bytes(s)) for s in states]
    for tag, net in nets:
        values, samp, sums, norms = [], [], [], []
        amax = 0.0
        for game, rng, ply in games:
            with torch.no_grad():
                v, p = net.forward(*features(ai, torch, game))
            lg = p.reshape(-1).numpy()
            values.append(float(v.item()))
            samp.append(lg[sidx].copy())
            sums.append(float(lg.astype(np.float64).sum()))
            norms.append(float(np.sqrt((lg.astype(np.float64) ** 2).sum())))
            amax = max(amax, float(np.abs(lg).max()))
        out["%s/values" % tag] = np.array(values, dtype=np.float32)
        out["%s/sampled_logits" % tag] = np.stack(samp).astype(np.float32)
        out["%s/logit_sum" % tag] = np.array(sums)
        out["%s/logit_l2" % tag] = np.array(norms)
        print("%-11s values %.4f..%.4f, |logit| max %.3f" % (tag, min(values), max(values), amax), flush=True)

    # ---- whole searches with the real net: MCTS(config, game, net) at MAX_ITER = 160, training = False -----------------
    # (no playout cap, no noise, argmax pick: the search is a function of the position, the net and the garbage draws,
    # which come from the engine generator through the harness as everywhere else)
    net = nets[0][1]
    cfg = ai.Config(visual=False, model="pytorch", model_config=AlphaSameConfig(blocks=10, filters=16), MAX_ITER=160,
                    training=False)
    order = np.argsort(-(hs[:, 0] + hs[:, 1]), kind="stable")
    sel = sorted(set(order[:N_SEARCH // 2].tolist()) | set(np.linspace(0, 63, 48).astype(int).tolist()))[:N_SEARCH]
    sel = sorted(sel)
    cases = []
    import time
    t0 = time.time()
    for i in sel:
        game, rng, ply = gg.game_from_state(bytes(states[i]))
        rh.install_rng(mods, rng)
        move, tree, save = ai.MCTS(cfg, game, net)
        root = tree.get_node("root")
        kids = [tree.get_node(c).data for c in root.successors(tree.identifier)]
        cases.append(dict(i=i, move=rh.encode_move(*move), moves=[rh.encode_move(*k.move) for k in kids],
                          visits=[int(k.visit_count) for k in kids], q=[float(k.value_avg) for k in kids],
                          p=[float(k.policy) for k in kids], garb=rng.garb_ctr))
        v = sorted(cases[-1]["visits"], reverse=True)
        print("search on position %2d: %3d children, move %5d, top visits %s  (%.0f s)" %
              (i, len(kids), cases[-1]["move"], v[:3], time.time() - t0), flush=True)
    out["search_pos"] = np.array([c["i"] for c in cases], dtype=np.int32)
    out["search_move"] = np.array([c["move"] for c in cases], dtype=np.int32)
    out["search_garb_ctr_after"] = np.array([c["garb"] for c in cases], dtype=np.int64)
    out["search_child_counts"] = np.array([len(c["moves"]) for c in cases], dtype=np.int32)
    out["search_child_moves"] = np.array([m for c in cases for m in c["moves"]], dtype=np.uint16)
    out["search_child_visits"] = np.array([v for c in cases for v in c["visits"]], dtype=np.int32)
    out["search_child_q"] = np.array([v for c in cases for v in c["q"]], dtype=np.float64)
    out["search_child_p"] = np.array([v for c in cases for v in c["p"]], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "net64.npz"), **out)
    print("wrote", os.path.join(OUT, "net64.npz"))


if __name__ == "__main__":
    main()
