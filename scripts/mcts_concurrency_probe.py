"""probe: what a round of ticks costs as more 256-slot groups tick side by side (1 .. 6 groups, each on its own stream): a
chip with room to spare would keep the round at one group's chain latency.
usage: python scripts/mcts_concurrency_probe.py [slots per group]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
from tetris_rl_amd import selfplay
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model='pytorch', model_config=mc, MAX_ITER=160, training=True)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
streams = selfplay.concurrent_streams(torch, torch.device("cuda", 0), 4)
streams = streams + [torch.cuda.Stream() for _ in range(2)]
parts = [trl.SelfPlay(cfg, net, M, seed=20260630 + g, record_capacity=65536, hip_stream=streams[g]) for g in range(6)]
for sp in parts:
    sp.tick(400)
for sp in parts:
    sp.synchronize()
for G in (1, 2, 3, 4, 5, 6):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for sp in parts[:G]: sp.tick(400)
        for sp in parts[:G]: sp.synchronize()
        best = min(best, (time.perf_counter() - t0) / 400)
    for sp in parts[G:]:
        sp.tick(1200)                                  # keep every group at the same depth of its games
    for sp in parts[G:]:
        sp.synchronize()
    print("%d groups of %d slots side by side: %.4f ms per round, %.1f slot-ticks per us" % (G, M, best * 1e3, G * M / (best * 1e3)), flush=True)
