"""probe: C3's 1024 slots ticking in G groups (SelfPlayGroups), G = 1, 2, 4, 8: ms per round of ticks"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model='pytorch', model_config=mc, MAX_ITER=160, training=True)
for groups in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    kw = dict(seed=20260630, use_graph=True, record_capacity=65536)
    sp = trl.SelfPlayGroups(cfg, net, 1024, groups, **kw) if groups > 1 else trl.SelfPlay(cfg, net, 1024, **kw)
    sp.tick(200); sp.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); sp.tick(800); sp.synchronize(); dt = time.perf_counter() - t0
        best = min(best, dt / 800)
    print('%d group(s) x %d slots: %.4f ms per round of ticks -> %.0f moves/s at 160 sims' % (groups, 1024 // groups, best * 1e3, 1024 / (160 * best)), flush=True)
    del sp
