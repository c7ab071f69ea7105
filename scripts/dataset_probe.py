"""probe: end-to-end make_training_set (device search -> records -> the reference's JSON set file), moves/s"""
import sys, time, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
os.environ["TRL_STORAGE"] = tempfile.mkdtemp()
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
net = trl.AlphaSame(mc)
import cProfile, pstats
for games in (256,):
    t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    data = trl.make_training_set(cfg, net, games, save_game=True, save_stats=True)
    pr.disable()
    dt = time.perf_counter() - t0
    size = os.path.getsize(os.path.join(cfg.data_dir, "0.txt"))
    print("%d games: %.1f s, set file %.0f MB" % (games, dt, size / 1e6), flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
