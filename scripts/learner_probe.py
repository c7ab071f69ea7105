"""probe: the learner's data path -- load_data_and_train_model (ai.py:1871-1921 mirror) over one self-play set of 64
games, AlphaSame(10,16), one epoch: from the packed set make_training_set writes, and from the JSON set file"""
import sys, time, os, tempfile, shutil, random
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
trl.make_training_set(cfg, net, 64, save_game=True, save_stats=True)
print("make_training_set(64 games, files written): %.1f s; set file %.0f MB, packed %.2f MB" % (
    time.perf_counter() - t0, os.path.getsize(cfg.data_dir + "/0.txt") / 1e6, os.path.getsize(learner.packed_dir(cfg) + "/0.npz") / 1e6), flush=True)
for packed in (True, False):
    if not packed:
        shutil.rmtree(learner.packed_dir(cfg))
    model = trl.AlphaSame(mc).cuda()
    random.seed(1); torch.manual_seed(1)
    t0 = time.perf_counter()
    entry = learner.load_data_and_train_model(cfg, model)
    torch.cuda.synchronize()
    print("load_data_and_train_model packed=%s: %.1f s, loss %.5f" % (packed, time.perf_counter() - t0, entry["loss"]), flush=True)
