"""Row (f3) on the device: the reference's training step (train_network_pytorch, ai.py:1139-1220) run on the MI355X
through PyTorch-ROCm -- losses within 1e-5 and every updated tensor within 1e-4 of the reference's CPU step
(tests/golden/learner.npz) -- and the multi-rank outer loop: games sharded, records to rank 0, learner on rank 0,
weights broadcast, gating match sharded; every rank ends with the same best network."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import tetris_rl_amd as trl
from test_learner import one_epoch_against_reference

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["alpha", "aux"])
def test_one_epoch_matches_reference_on_device(golden_dir, tag, tmp_path, monkeypatch):
    monkeypatch.setenv("TRL_STORAGE", str(tmp_path))
    one_epoch_against_reference(golden_dir, tag, "cuda")


WORKER = r'''
import hashlib, json, os, random, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import tetris_rl_amd as trl
dist.init_process_group("gloo")
random.seed(0); torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=2, filters=8)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=8, training=False, training_games=4, battle_games=4,
                 sets_to_train_with=2, batch_size=16)
n = trl.self_play_loop(cfg, max_sets=2, seed=1, dist=dist)
best = trl.load_best_model(cfg) if dist.get_rank() == 0 else None
# the network every rank holds after the loop = what rank 0 stored (when something was promoted)
from tetris_rl_amd import selfplay
h = hashlib.sha256()
net = selfplay._last_best
for k, v in net.state_dict().items():
    h.update(v.detach().cpu().numpy().tobytes())
json.dump({"rank": dist.get_rank(), "sets": n, "weights": h.hexdigest()}, open(%(out)r %% dist.get_rank(), "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_self_play_loop(tmp_path):
    """two ranks on this box's GPU (gloo): both play half of each set and of each gating match; rank 0 writes Storage/;
    after two rounds both ranks hold bit-identical weights, and set 0 is byte-identical to the single-process run's"""
    out = str(tmp_path / "rank%d.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=out))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TRL_STORAGE=str(tmp_path / "Storage2"))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], check=True, env=env, timeout=900)
    r0, r1 = (json.load(open(out % r)) for r in (0, 1))
    assert r0["sets"] == r1["sets"] == 2 and r0["weights"] == r1["weights"]
    store2 = tmp_path / "Storage2"
    gl = [json.loads(l) for l in open(store2 / "logs" / "gating_log.jsonl")]
    assert len(gl) == 2 and len(open(store2 / "logs" / "training_log.jsonl").read().splitlines()) == 2
    # single process, same seeds: the first set (played by the same initial network) is the same file
    import random
    os.environ["TRL_STORAGE"] = str(tmp_path / "Storage1")
    try:
        random.seed(0); torch.manual_seed(0)
        mc = trl.AlphaSameConfig(blocks=2, filters=8)
        cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=8, training=False, training_games=4,
                         battle_games=4, sets_to_train_with=2, batch_size=16)
        trl.self_play_loop(cfg, max_sets=1, seed=1)
        a = open(os.path.join(cfg.data_dir, "0.txt"), "rb").read()
    finally:
        os.environ.pop("TRL_STORAGE")
    b = open(str(store2 / "data" / ("%s.%s" % (cfg.ruleset, cfg.data_version)) / "0.txt"), "rb").read()
    assert a == b and len(a) > 1000
