"""C4's exchange step on CPU: two ranks hold the device-format records of different games, gather them
to rank 0 over torch.distributed (gloo here, RCCL on the GPUs) and rank 0 rebuilds the reference's
13-field records -- which must hash to the reference's play_game output (tests/golden/selfplay.npz)."""
import hashlib
import json
import os
import socket

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import tetris_rl_amd as trl
from tetris_rl_amd import parallel

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _records_of_game(d, gi, winner):
    """device-format records (RECORD_DTYPE) of fixture game gi"""
    p_off = np.concatenate([[0], np.cumsum(d["game_plies"])])
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    training = bool(d["game_training"][gi])
    out = np.zeros(int(d["game_plies"][gi]) + 1, dtype=trl.RECORD_DTYPE)
    for j, k in enumerate(range(p_off[gi], p_off[gi + 1])):
        r = out[j]
        n = int(d["ply_child_counts"][k])
        r["slot"], r["episode"], r["ply"], r["n_children"], r["kind"] = gi, 0, j, n, trl.REC_MOVE
        r["chosen"], r["saved"] = d["ply_move"][k], 1 if (d["ply_save"][k] or not training) else 0
        r["state"] = d["ply_state"][k]
        r["moves"][:n] = d["ply_child_moves"][c_off[k]:c_off[k + 1]]
        r["visits"][:n] = d["ply_child_visits"][c_off[k]:c_off[k + 1]]
    e = out[-1]
    e["slot"], e["episode"], e["ply"], e["kind"], e["winner"] = gi, 0, int(d["game_plies"][gi]), trl.REC_GAME_END, winner
    return out


def _winner_from_z(d, gi):
    r_off = np.concatenate([[0], np.cumsum(d["game_records"])])
    z0 = float(d["rec_z"][r_off[gi]])          # first record belongs to player 0
    return -1 if z0 == 0.5 else (0 if z0 == 1.0 else 1)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(GOLDEN, "selfplay.npz"))
    mine = [0] if rank == 0 else [1]                     # the two training games
    recs = np.concatenate([_records_of_game(d, gi, _winner_from_z(d, gi)) for gi in mine])
    parts = parallel.gather_bytes(np.frombuffer(recs.tobytes(), dtype=np.uint8), dist)
    if rank == 0:
        allrecs = np.concatenate([np.frombuffer(p.tobytes(), dtype=trl.RECORD_DTYPE) for p in parts])
        games = trl.games_from_records(trl.Config(training=True), allrecs)
        q.put([[hashlib.sha256(json.dumps(r, separators=(",", ":")).encode()).hexdigest() for r in g["data"]] for g in games])
    dist.barrier()
    dist.destroy_process_group()


def test_gathered_records_rebuild_reference_games():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    hashes = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = np.load(os.path.join(GOLDEN, "selfplay.npz"))
    r_off = np.concatenate([[0], np.cumsum(d["game_records"])])
    assert len(hashes) == 2
    for gi in (0, 1):
        assert hashes[gi] == [str(h) for h in d["rec_hash"][r_off[gi]:r_off[gi + 1]]]
