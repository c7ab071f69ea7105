"""probe: what bounds the C3 round of ticks with G slot groups -- the same run with the tower launch left out (search
kernels only) and with the search launch left out (towers only; every board flagged).  Timing only, the games are garbage."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
from tetris_rl_amd import selfplay
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model='pytorch', model_config=mc, MAX_ITER=160, training=True)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for mode in ("both", "search only", "tower only (half the boards flagged)"):
    streams = selfplay.concurrent_streams(torch, torch.device("cuda", 0), G)
    parts = []
    for g in range(G):
        sp = trl.SelfPlay(cfg, net, 1024 // G, seed=20260630 + g, use_graph=False, record_capacity=65536, hip_stream=streams[g])
        if mode == "search only":
            sp.net = type("N", (), {"tower_need": staticmethod(lambda *a, **k: None), "tower_mode": "split"})()
        elif mode.startswith("tower"):
            with torch.cuda.stream(sp.stream):
                sp.need[sp.n:] = 1; sp.need[:sp.n] = 0           # the steady state: one board per leaf
            sp.engine.mcts_expand_alphasame_select = lambda *a, **k: None
            sp.engine.mcts_select_need = lambda *a, **k: None
        sp._capture()
        parts.append(sp)
    for sp in parts: sp.tick(100)
    for sp in parts: sp.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for sp in parts: sp.tick(800)
        for sp in parts: sp.synchronize()
        best = min(best, (time.perf_counter() - t0) / 800)
    print("%d groups, %-40s %.4f ms per round" % (G, mode + ":", best * 1e3), flush=True)
    del parts
