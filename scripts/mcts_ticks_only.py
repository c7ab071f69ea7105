"""probe for profiler passes: 300 search ticks of one C3 slot group (256 slots), nothing else"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256        # slots of one group of the production tick (C3: 4 groups x 256)
sp = trl.SelfPlay(cfg, trl.AlphaSame(mc), n, seed=20260630, use_graph=True, record_capacity=65536)
sp.tick(300); sp.synchronize()
print("ticks done")
