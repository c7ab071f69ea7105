import sys, time, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TRL_STORAGE", tempfile.mkdtemp())
import numpy as np, torch
import tetris_rl_amd as trl
torch.manual_seed(0)
for mc, games, it in ((trl.AlphaSameConfig(blocks=10, filters=16), 12288, 160), (trl.AlphaSameConfig(blocks=4, filters=64), 2048, 160)):
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=it, training=True)
    net = trl.AlphaSame(mc)
    t0 = time.perf_counter()
    trl.make_training_set(cfg, net, games, save_game=True, save_stats=True, seed=9, write_json=False)
    dt = time.perf_counter() - t0
    print("filters %d: %d games in %.1f s -> %.0f games/s" % (mc.filters, games, dt, games / dt), flush=True)
