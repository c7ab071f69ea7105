"""probe: train_network_pytorch (ai.py:1139-1220 mirror) on one self-play set, GPU; AlphaSame(10,16)"""
import sys, time, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TRL_STORAGE"] = tempfile.mkdtemp()
import torch
import tetris_rl_amd as trl
from tetris_rl_amd import learner
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True, epochs=1)
net = trl.AlphaSame(mc)
t0 = time.perf_counter()
data = trl.make_training_set(cfg, net, 64, save_game=False, save_stats=False)
print("set: %d records in %.1f s" % (len(data), time.perf_counter() - t0), flush=True)
model = trl.AlphaSame(mc).cuda()
t0 = time.perf_counter()
entry = learner.train_network_pytorch(cfg, model, data)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("train_network_pytorch: %d records, 1 epoch, batch %d: %.1f s (%.0f records/s), loss %.4f" % (len(data), cfg.batch_size, dt, len(data) / dt, entry["loss"]))
