"""probe for profiler passes: 300 C3 search ticks, nothing else"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
sp = trl.SelfPlay(cfg, trl.AlphaSame(mc), 1024, seed=20260630, use_graph=True, record_capacity=65536)
sp.tick(300); sp.synchronize()
print("ticks done")
