"""soak / end-to-end rate: 3072 self-play games on 1024 slots (auto-reset, several episodes per slot), AlphaSame(10,16),
MAX_ITER=160, training=True, through make_training_set with the packed set only (write_json=False) -- every game must
end, no capacity may overflow"""
import sys, time, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TRL_STORAGE", tempfile.mkdtemp())
import numpy as np, torch
import tetris_rl_amd as trl
from tetris_rl_amd import learner
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
net = trl.AlphaSame(mc)
for write_json, games in ((False, 3072), (True, 512)):
    t0 = time.perf_counter()
    trl.make_training_set(cfg, net, games, save_game=True, save_stats=True, seed=4, write_json=write_json)
    dt = time.perf_counter() - t0
    n = trl.highest_data_number(cfg)
    ps = np.load("%s/%d.npz" % (learner.packed_dir(cfg), n))
    plies = None
    print("write_json=%s: %d games, %d saved moves (%d records) in %.1f s -> %.0f games/s" % (
        write_json, games, len(ps["nchild"]), 4 * len(ps["nchild"]), dt, games / dt), flush=True)
