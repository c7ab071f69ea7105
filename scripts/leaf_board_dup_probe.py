"""probe: how many of the boards the tower evaluates in a search have been evaluated before -- in the same search (the
mover's board after a move depends on that player's board and the move only, so the same board comes up under different
replies of the other player) or in the slot's previous searches.   usage: python scripts/leaf_board_dup_probe.py [slots] [ticks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import tetris_rl_amd as trl
slots = int(sys.argv[1]) if len(sys.argv) > 1 else 96
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 700
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # ticks played before counting (later game phases)
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
os.environ["TRL_LEGAL_IN_TOWER"] = "0"
sp = trl.SelfPlay(cfg, net, slots, seed=20260630, auto_reset=True, use_graph=True)
if warm:
    sp.tick(warm); sp.synchronize()
seen_search = [set() for _ in range(slots)]       # boards of the current search
seen_slot = [set() for _ in range(slots)]         # boards of the slot's earlier searches of this game
last_iter = np.zeros(slots, dtype=np.int64)
tot = dup_search = dup_prev = 0
by_iter = {}
for t in range(ticks):
    sp.tick(1)
    sp.synchronize()
    states, info = sp.engine.mcts_leaf_states()     # the pending selection's leaves
    for s in range(slots):
        it, mx, kind = int(info[s, 0]), int(info[s, 1]), int(info[s, 2])
        if kind > 1:
            continue
        if it <= last_iter[s]:                      # a new search began
            seen_slot[s] |= seen_search[s]
            if len(seen_slot[s]) > 4000:
                seen_slot[s] = set()
            seen_search[s] = set()
        last_iter[s] = it
        st = states[s]
        turn = int(st[312])
        mover = 1 - turn
        key = (mover, bytes(st[144 * mover:144 * mover + 80]))
        if it == 1:
            continue                                # the root: both boards evaluated, not a lookup candidate here
        tot += 1
        b = by_iter.setdefault(min(it // 40, 9), [0, 0])
        b[0] += 1
        if key in seen_search[s]:
            dup_search += 1
            b[1] += 1
        elif key in seen_slot[s]:
            dup_prev += 1
        seen_search[s].add(key)
print("after %d ticks of play: " % warm, end="")
print("%d leaf evaluations: %.1f %% of the mover's boards were evaluated earlier in the same search, another %.1f %% in an "
      "earlier search of the slot" % (tot, 100.0 * dup_search / tot, 100.0 * dup_prev / tot))
for k in sorted(by_iter):
    print("  simulations %3d..%3d: %5d leaves, %.1f %% seen in the same search" % (40 * k, 40 * k + 39, by_iter[k][0], 100.0 * by_iter[k][1] / max(by_iter[k][0], 1)))
