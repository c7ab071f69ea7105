"""C4 on the device: make_training_set sharded over two ranks (one process per rank, both on this box's GPU, gloo
for the record gather -- RCCL refuses two ranks on one device) must return exactly the games of the single-process
run: rank r plays games [r*G/W, (r+1)*G/W) on RNG streams of the same numbers."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

import tetris_rl_amd as trl

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
import tetris_rl_amd as trl
dist.init_process_group("gloo")
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=2, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=12, training=True)
net = trl.AlphaSame(mc)
data = trl.make_training_set(cfg, net, 8, save_stats=False, seed=77, dist=dist)
if dist.get_rank() == 0:
    json.dump(data, open(%(out)r, "w"))
else:
    assert data is None
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_training_set_equals_single_process(tmp_path):
    out = str(tmp_path / "dist.json")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, out=out))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, TRL_STORAGE=str(tmp_path / "Storage"))
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                    "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], check=True, env=env, timeout=600)
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=12, training=True)
    single = trl.make_training_set(cfg, trl.AlphaSame(mc), 8, save_stats=False, seed=77)
    both = json.load(open(out))
    assert len(both) == len(single) > 0
    assert json.dumps(both) == json.dumps(single)
