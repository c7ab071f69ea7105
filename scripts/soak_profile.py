"""probe: where the host spends its time in make_training_set with the packed set only (write_json=False), 3072 games on 1024
slots -- cProfile, cumulative"""
import sys, time, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TRL_STORAGE", tempfile.mkdtemp())
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
net = trl.AlphaSame(mc)
trl.make_training_set(cfg, net, 64, save_game=True, save_stats=True, seed=3, write_json=False)      # warm-up (build, allocations)
import cProfile, pstats
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
trl.make_training_set(cfg, net, 3072, save_game=True, save_stats=True, seed=4, write_json=False)
pr.disable()
print("3072 games in %.2f s" % (time.perf_counter() - t0), flush=True)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
