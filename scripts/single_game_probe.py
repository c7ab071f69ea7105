"""probe: BASELINE config C1 -- one self-play game, ruleset s2, AlphaSame(10,16), MAX_ITER=160 (the reference's
simulation.py case; 0.77 moves/s in Python on one core) -- through play_game on the device (1 slot)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, tempfile
os.environ.setdefault("TRL_STORAGE", tempfile.mkdtemp())
import tetris_rl_amd as trl
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=False)
net = trl.AlphaSame(mc)
trl.play_game(cfg, net, game_number=1)              # warm-up (library load, graph capture)
t0 = time.perf_counter()
data, stats = trl.play_game(cfg, net, game_number=2)
dt = time.perf_counter() - t0
plies = len(data) // 4
print("one game: %d saved positions in %.2f s -> %.1f moves/s (160 simulations each)" % (plies, dt, plies / dt))
