"""Multi-GPU layout of the rollout path: one process per GPU, games sharded by rank.

Self-play games are independent (ai.py:1817-1820, 1854-1860), so the env / search path has no
collective; torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used
for the barrier, for timing reductions and for gathering training records to rank 0.
"""
import os

import numpy as np


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1 process: 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def game_streams(rank, games_per_rank, episode_base=0):
    """RNG stream ids of the games owned by `rank`: global game id = rank * G + g.
    Streams of different ranks are disjoint, and the union over ranks of a W-rank job equals
    the streams of a single-rank job with W * G games."""
    g = np.arange(games_per_rank, dtype=np.uint64) + np.uint64(rank) * np.uint64(games_per_rank)
    if (g >= (1 << 24)).any():
        raise ValueError("more than 2^24 concurrent games: stream ids would collide with auto-reset episodes")
    return (g + np.uint64(episode_base)).astype(np.uint32)


def reduce_job(elapsed_s, units, dist=None, device=None):
    """Whole-job numbers from per-rank ones: max elapsed time, total units processed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), int(units)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([int(units)], dtype=torch.int64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def gather_bytes(payload, dist=None, device=None, dst=0):
    """Gather one variable-length uint8 array per rank to `dst` (all_gather of lengths, then a
    padded gather) -- the exchange step for packed training records.  Returns the list of
    per-rank arrays on `dst`, None elsewhere."""
    payload = np.ascontiguousarray(payload, dtype=np.uint8).reshape(-1)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [payload]
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([payload.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    if payload.size:
        buf[:payload.size] = torch.from_numpy(payload).to(buf.device)
    # all_gather is used instead of gather: RCCL and gloo both implement it for every dtype
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    if rank != dst:
        return None
    return [o[:s].cpu().numpy() for o, s in zip(out, sizes)]
