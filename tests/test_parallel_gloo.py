"""world_size-2 gloo test of the multi-GPU layout: game -> rank sharding, job reductions and the
record gather, on CPU."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tetris_rl_amd import parallel


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert parallel.dist_env() == (rank, rank, world)
    streams = parallel.game_streams(rank, 8)
    elapsed, units = parallel.reduce_job(1.0 + rank, 100 * (rank + 1), dist)
    payload = np.arange(5 + 3 * rank, dtype=np.uint8) + rank
    got = parallel.gather_bytes(payload, dist)
    # weight broadcast of the outer loop: every rank ends with rank 0's parameters and buffers
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    net[1].num_batches_tracked.fill_(7 + rank)
    parallel.broadcast_state_dict(net, dist)
    digest = [float(v.double().sum()) for v in net.state_dict().values()]
    ints = parallel.all_gather_ints([rank, 10 * rank + 1], dist).tolist()
    # host-side rendezvous in front of a collective (self_play_loop: rank 0 trains while the others wait): rank 1 returns
    # from the wait only after rank 0 has arrived
    import time
    if rank == 0:
        time.sleep(1.0)
    t0 = time.perf_counter()
    parallel.wait_for_rank0(dist)
    waited = time.perf_counter() - t0
    assert rank == 0 or waited > 0.5, waited
    q.put((rank, streams.tolist(), elapsed, units, None if got is None else [g.tolist() for g in got], digest, ints))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    s0, s1 = res[0][1], res[1][1]
    assert not set(s0) & set(s1)
    assert sorted(s0 + s1) == parallel.game_streams(0, 16).tolist()      # sharded == unsharded
    for r in res:
        assert r[2] == 2.0 and r[3] == 300                                # max time, total units
    assert res[1][4] is None
    assert res[0][4] == [list(range(5)), [1 + i for i in range(8)]]
    assert res[0][5] == res[1][5] and res[0][5][-1] == 7.0               # rank 0's weights (and its BN counter) everywhere
    assert res[0][6] == res[1][6] == [[0, 1], [1, 11]]


def test_single_process_paths():
    assert parallel.reduce_job(0.5, 7) == (0.5, 7)
    assert [a.tolist() for a in parallel.gather_bytes(np.array([1, 2, 3], dtype=np.uint8))] == [[1, 2, 3]]
    assert parallel.game_streams(3, 4).tolist() == [12, 13, 14, 15]
