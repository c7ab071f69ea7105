"""probe: ms per search tick and its split (not a test, not the bench)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
torch.manual_seed(0)
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16))
cfg = trl.Config(visual=False, model='pytorch', model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=160, training=True)
for n in (1024,):
    for hip in (True, False):
        net.use_hip_trunk = hip
        sp = trl.SelfPlay(cfg, net, n, seed=20260630, use_graph=True, record_capacity=65536)
        sp.tick(20); sp.synchronize()
        t0=time.perf_counter(); sp.tick(300); sp.synchronize(); dt=time.perf_counter()-t0
        print(n, 'slots hip_trunk=%s: %.3f ms/tick -> %.0f moves/s at 160 sims' % (hip, dt/300*1e3, n/(160*dt/300)), flush=True)
        with torch.cuda.stream(sp.stream):
            torch.cuda.synchronize()
            for name, fn in (('net', lambda: sp.net.evaluate_packed(sp.grids, sp.vec)), ('trunk', lambda: sp.net.trunk(sp.grids.reshape(2*n,1,40,10)))):
                fn(); torch.cuda.synchronize(); t0=time.perf_counter()
                for _ in range(50): fn()
                torch.cuda.synchronize(); print('   %s %.3f ms' % (name, (time.perf_counter()-t0)/50*1e3), flush=True)
        del sp
