"""Multi-GPU layout of the rollout path: one process per GPU, games sharded by rank.

Self-play games are independent (ai.py:1817-1820, 1854-1860), so the env / search path has no
collective; torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests) is used
for the barrier, for timing reductions, for gathering training records to rank 0 and for
broadcasting promoted weights.
"""
import os

import numpy as np


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1 process: 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def active(dist):
    """a process group with more than one rank -- or, with TRL_DIST_SINGLE_RANK=1, an initialised group of one: the
    rehearsal of the multi-rank code path (RCCL calls on device tensors included) on a box with one GPU"""
    if dist is None or not dist.is_initialized():
        return False
    return dist.get_world_size() > 1 or os.environ.get("TRL_DIST_SINGLE_RANK", "0") == "1"


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
    if not active(dist):
        return float(elapsed_s), int(units)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([int(units)], dtype=torch.int64, device=device)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def all_gather_ints(values, dist, device=None):
    """every rank's small tuple of ints on every rank: int64 [world, len(values)] as a numpy array"""
    import torch
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    out = torch.empty(dist.get_world_size() * mine.numel(), dtype=torch.int64, device=device)     # flat: gloo insists
    dist.all_gather_into_tensor(out, mine)
    return out.cpu().numpy().reshape(dist.get_world_size(), mine.numel())


def gather_tensor(payload, sizes, dist, dst=0):
    """The exchange step of the path: one variable-length uint8 tensor per rank -> list of per-rank tensors on `dst`,
    None elsewhere.  `payload` lives where the backend works (HBM for RCCL: the records go GPU -> GPU over xGMI and
    never touch the sender's host; host memory for gloo); `sizes[r]` = bytes of rank r (from all_gather_ints).  A padded
    dist.gather: only `dst` receives anything."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = int(max(max(sizes), 1))
    if payload.numel() == cap:
        buf = payload
    else:
        buf = torch.empty(cap, dtype=torch.uint8, device=payload.device)
        buf[:payload.numel()] = payload
    out = [torch.empty(cap, dtype=torch.uint8, device=payload.device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [o[:int(s)] for o, s in zip(out, sizes)]


def gather_bytes(payload, dist=None, device=None, dst=0):
    """gather_tensor for a numpy uint8 array per rank (lengths exchanged first).  Returns the list of per-rank
    arrays on `dst`, None elsewhere."""
    payload = np.ascontiguousarray(payload, dtype=np.uint8).reshape(-1)
    if not active(dist):
        return [payload]
    import torch
    sizes = all_gather_ints([payload.size], dist, device)[:, 0]
    t = torch.from_numpy(payload)
    if device is not None:
        t = t.to(device)
    parts = gather_tensor(t, sizes, dist, dst)
    if parts is None:
        return None
    return [p.cpu().numpy() for p in parts]


_wait_seq = 0


def wait_for_rank0(dist, hours=48.0):
    """Host-side rendezvous in front of a collective that the other ranks would otherwise sit in while rank 0 works
    alone (training the challenger): a rank parked INSIDE an RCCL / gloo collective is subject to the process group's
    timeout (10 minutes by default) and takes the job down when rank 0 needs longer.  Rank 0 sets a key in the group's
    store when it arrives, the others wait for it there -- the store's wait takes its own, long timeout.  Every rank
    calls this the same number of times; no-op without a group."""
    global _wait_seq
    if not active(dist):
        return
    from datetime import timedelta
    _wait_seq += 1
    key = "trl_rank0_ready_%d" % _wait_seq
    try:
        store = dist.distributed_c10d._get_default_store()
    except Exception:
        return                                    # no store to wait on: the collective itself is the rendezvous
    if dist.get_rank() == 0:
        store.set(key, "1")
    else:
        store.wait([key], timedelta(hours=hours))


def broadcast_state_dict(module, dist, src=0):
    """rank `src`'s parameters and buffers to every rank, in place (the refreshed weights of the outer loop,
    SURVEY.md section 8e: one flat bucket, one broadcast -- 24 MB for AlphaSame(10, 16))"""
    if not active(dist):
        return module
    import torch
    tensors = [t for t in module.state_dict().values() if t.is_floating_point()]
    others = [t for t in module.state_dict().values() if not t.is_floating_point()]
    host = dist.get_backend() != "nccl"           # gloo moves host memory
    if tensors:
        flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
        if host:
            flat = flat.cpu()
        dist.broadcast(flat, src=src)
        off = 0
        with torch.no_grad():
            for t in tensors:
                n = t.numel()
                t.copy_(flat[off:off + n].reshape(t.shape).to(t.dtype))
                off += n
    for t in others:                         # num_batches_tracked counters
        b = t.cpu() if host else t
        dist.broadcast(b, src=src)
        if b is not t:
            t.copy_(b)
    return module
