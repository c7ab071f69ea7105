"""probe: does splitting the slots into independently ticking groups (own stream + graph each) overlap
the search kernels of one group with the tower of another?"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
from tetris_rl_amd import parallel, engine
if len(sys.argv) > 1:                                   # another build of the library (A/B in one box)
    engine._lib = engine.load_library(sys.argv[1])
torch.manual_seed(0)
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16))
cfg = trl.Config(visual=False, model='pytorch', model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=160, training=True)
for total, groups in ((1024, 1), (1024, 2), (2048, 1), (2048, 2), (3072, 3), (4096, 4))[:2 if len(sys.argv) > 1 else 6]:
    n = total // groups
    sps = [trl.SelfPlay(cfg, net, n, seed=20260630, streams=parallel.game_streams(g, n), use_graph=True, record_capacity=65536) for g in range(groups)]
    for sp in sps: sp.tick(20)
    for sp in sps: sp.synchronize()
    T = 300
    t0 = time.perf_counter()
    for _ in range(T):
        for sp in sps: sp.tick(1)
    for sp in sps: sp.synchronize()
    dt = time.perf_counter() - t0
    print('%d group(s) x %d slots: %.3f ms per round of ticks -> %.0f moves/s at 160 sims' % (groups, n, dt / T * 1e3, total / (160 * dt / T)), flush=True)
    del sps
