import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import tetris_rl_amd as trl
torch.manual_seed(0)
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16))
cfg = trl.Config(visual=False, model='pytorch', model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=160, training=True)
for n in (256, 1024):
    sp = trl.SelfPlay(cfg, net, n, seed=20260630, use_graph=True, record_capacity=65536)
    sp.tick(20); sp.synchronize()
    t0=time.perf_counter(); sp.tick(200); sp.synchronize(); dt=time.perf_counter()-t0
    st = sp.stats()
    print(n, 'slots: %.3f ms/tick' % (dt/200*1e3), st, flush=True)
    # breakdown: net only
    with torch.cuda.stream(sp.stream):
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(50): sp.net.evaluate_packed(sp.grids, sp.vec)
        torch.cuda.synchronize(); print('  net eager %.3f ms' % ((time.perf_counter()-t0)/50*1e3))
        t0=time.perf_counter()
        for _ in range(50): sp.net.trunk(sp.grids.reshape(2*n,1,40,10))
        torch.cuda.synchronize(); print('  trunk eager %.3f ms' % ((time.perf_counter()-t0)/50*1e3))
    del sp
