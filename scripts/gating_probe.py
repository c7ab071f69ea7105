"""probe: a gating match (battle_networks) of two AlphaSame(10,16) nets, 256 concurrent games, MAX_ITER=160:
native two-network tick against the generic evaluator-callable tick"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import tetris_rl_amd as trl
from tetris_rl_amd import gating
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=False)
torch.manual_seed(0); a = trl.AlphaSame(mc).cuda().prepare()
torch.manual_seed(1); b = trl.AlphaSame(mc).cuda().prepare()
games = 256
for native in (True, False):
    bt = gating.Battle(cfg, a.evaluate_packed, b.evaluate_packed, games, seed=5, nets=(a, b) if native else None)
    bt.tick(10); bt.stream.synchronize()
    t0 = time.perf_counter(); bt.tick(100); bt.stream.synchronize(); dt = (time.perf_counter() - t0) / 100
    print("native=%s: %.3f ms/tick (%d games)" % (native, dt * 1e3, games), flush=True)
t0 = time.perf_counter()
wins, acc = gating.battle_networks(a, cfg, b, cfg, 0.55, "more", games, seed=5)
print("battle_networks: %d games in %.1f s, wins %s accepted %s" % (games, time.perf_counter() - t0, wins, acc))
