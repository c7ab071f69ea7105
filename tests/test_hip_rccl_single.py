"""The RCCL branches of parallel.py (device tensors through torch.distributed's "nccl" backend) on the one GPU this box has:
a process group of ONE rank -- no second device to talk to, but the same calls, tensor placements, dtypes and shapes the
N-GPU job makes: all_gather_into_tensor of the int64 size tuple, the padded uint8 dist.gather of the records from HBM,
all_reduce MAX / SUM of the timing reduction, the flat fp32 broadcast of the weights -- and make_training_set's multi-rank
path as a whole (TRL_DIST_SINGLE_RANK=1).  (Two ranks on one device are refused by RCCL; the two-rank tests of this suite
run over gloo.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch, torch.distributed as dist
import tetris_rl_amd as trl
from tetris_rl_amd import parallel
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
assert dist.get_backend() == "nccl"
info = parallel.all_gather_ints([12345, 1, 0], dist, dev)
assert info.shape == (1, 3) and info[0].tolist() == [12345, 1, 0]
payload = (torch.arange(2392 * 7, device=dev) %% 251).to(torch.uint8)
parts = parallel.gather_tensor(payload, [payload.numel()], dist)
assert len(parts) == 1 and parts[0].is_cuda and torch.equal(parts[0], payload)
t = torch.tensor([1.25], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([77], dtype=torch.int64, device=dev); dist.all_reduce(u, op=dist.ReduceOp.SUM)
assert float(t.item()) == 1.25 and int(u.item()) == 77
torch.manual_seed(0)
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=16)).cuda()
before = {k: v.clone() for k, v in net.state_dict().items()}
flat = torch.cat([v.detach().reshape(-1).to(torch.float32) for v in net.state_dict().values() if v.is_floating_point()])
dist.broadcast(flat, src=0)                                   # the one-bucket weight broadcast of the outer loop
assert flat.is_cuda and all(torch.equal(before[k], v) for k, v in net.state_dict().items())
# the self-play driver's multi-rank path itself (records drained on the device, sizes all-gathered, payload gathered over
# the nccl backend, converted on rank 0) with this one rank: the same training set as the single-process path
import json
os.environ["TRL_DIST_SINGLE_RANK"] = "1"
assert parallel.active(dist)
mc = trl.AlphaSameConfig(blocks=2, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=12, training=True)
torch.manual_seed(0)
multi = trl.make_training_set(cfg, trl.AlphaSame(mc), 8, save_stats=False, seed=77, dist=dist)
os.environ["TRL_DIST_SINGLE_RANK"] = "0"
torch.manual_seed(0)
single = trl.make_training_set(cfg, trl.AlphaSame(mc), 8, save_stats=False, seed=77)
assert len(multi) == len(single) > 0 and json.dumps(multi) == json.dumps(single)
dist.barrier()
dist.destroy_process_group()
print("rccl single-rank ok")
'''


def test_rccl_calls_of_the_multi_gpu_path_with_one_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, timeout=600, capture_output=True, text=True)
    assert r.returncode == 0 and "rccl single-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_line_through_the_rccl_path_with_one_rank(tmp_path):
    """bench.py with a process group of its one rank (TRL_DIST_SINGLE_RANK=1): barrier, the job reductions, C3's record
    gather from HBM and the per-rank rates all go through the nccl backend; the line says so"""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRL_DIST_SINGLE_RANK="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--games", "256",
                        "--launches", "3", "--mcts-games", "64", "--mcts-ticks", "96", "--mcts-warmup", "16", "--no-c5", "--no-c1",
                        "--no-cpu-baseline"], env=env, timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["dist"]["backend"] == "nccl" and line["dist"]["world_size"] == 1
    assert line["value"] > 0 and line["mcts"]["value"] > 0 and line["mcts"]["per_rank_value"] and len(line["mcts"]["per_rank_value"]) == 1
    assert line["mcts"]["config"]["records_to_rank0_mb"] > 0
