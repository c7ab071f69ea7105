"""probe: duration of the tick's two launches of one 256-slot group (others idle), eager, HIP events: median / mean / p90 --
what the legal-move job adds to the tower launch.   usage: [TRL_ALT_LIB=...] python scripts/tower_launch_probe.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
if os.environ.get("TRL_ALT_LIB"):
    from tetris_rl_amd import engine
    engine._lib = engine.load_library(os.environ["TRL_ALT_LIB"])
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
sp = trl.SelfPlay(cfg, trl.AlphaSame(mc), 256, seed=20260630, use_graph=True, record_capacity=65536)
sp.tick(600); sp.synchronize()
reps = 400
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
with torch.cuda.stream(sp.stream):
    for e0, e1, e1b, e2 in ev:
        e0.record(sp.stream); sp._tower(); e1.record(sp.stream); e1b.record(sp.stream)
        sp.engine.mcts_expand_alphasame_select(sp.z.data_ptr(), sp.vec.data_ptr(), sp.heads, sp.grids.data_ptr(), sp.need.data_ptr())
        e2.record(sp.stream)
sp.synchronize()
over = statistics.median(e1.elapsed_time(e1b) for e0, e1, e1b, e2 in ev)
t = np.array([e0.elapsed_time(e1) - over for e0, e1, e1b, e2 in ev[20:]]) * 1e3
s = np.array([e1b.elapsed_time(e2) - over for e0, e1, e1b, e2 in ev[20:]]) * 1e3
print("%s: tower launch median %.1f mean %.1f p90 %.1f us | search median %.1f mean %.1f p90 %.1f us (legal in tower: %s)" %
      (os.environ.get("TRL_ALT_LIB", "default build"), np.median(t), t.mean(), np.percentile(t, 90), np.median(s), s.mean(), np.percentile(s, 90), sp.legal_in_tower))
