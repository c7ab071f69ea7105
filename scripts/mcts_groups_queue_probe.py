"""probe: is the bimodal 4-group tick time (0.16 / 0.25 ms) a collision of the groups' streams on one hardware queue?
runs the 4-group configuration several times in one process, each time after creating k dummy streams (which shifts the
groups' streams round the hardware queues)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model='pytorch', model_config=mc, MAX_ITER=160, training=True)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
keep = []
for trial in range(6):
    groups = 4
    sp = trl.make_self_play(cfg, net, 1024, seed=20260630, use_graph=True, record_capacity=65536)
    assert len(sp.parts) == 4
    sp.tick(200); sp.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); sp.tick(800); sp.synchronize(); dt = time.perf_counter() - t0
        best = min(best, dt / 800)
    print('trial %d: streams %s: %.4f ms per round -> %.0f moves/s' % (trial, [hex(p.stream.cuda_stream)[-6:] for p in sp.parts], best * 1e3, 1024 / (160 * best)), flush=True)
    del sp
    keep.append(torch.cuda.Stream())          # shift the pool position by one
