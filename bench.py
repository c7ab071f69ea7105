#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused HIP env-step kernel on BASELINE.json's C2 workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one ply of every game of the batch: legal-placement enumeration for the side to
move, a uniform pick among the placements, Game.make_move (lock, line clear, attack, garbage
exchange, spawn, bag refill), and Game.setup() again for games that ended.  The K timed steps
run as one launch of the rollout kernel (the state of a game stays in LDS between plies);
W warm-up steps run before.  Inputs are synthetic (per-game counter-based RNG streams,
seed 20260630) and resident in HBM before the timed region.

Multi-GPU: games are independent, so each rank owns its own 4096 games with disjoint RNG
streams (weak scaling) and there is no data-path collective; torch.distributed (RCCL) only
provides the barrier and the max-over-ranks reduction of the elapsed time.

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20260630
GAMES_PER_GPU = 4096
BYTES_PER_STEP = 320          # algorithmic bytes of one env-step
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(seconds_target=15.0):
    """The oracle's statement of the same workload on one host core (baseline only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    games, plies = 64, 200
    ob = op.OracleBatch(games, ruleset=2, seed=SEED)
    t0 = time.perf_counter()
    steps = ob.rollout_random(plies)
    dt = time.perf_counter() - t0
    rate = steps / dt
    # scale the sample to ~seconds_target of CPU work
    games2 = max(64, min(4096, int(rate * seconds_target / 1000)))
    ob = op.OracleBatch(games2, ruleset=2, seed=SEED)
    t0 = time.perf_counter()
    steps = ob.rollout_random(1000)
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 1), "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": "oracle/tetris_oracle.c, games 0..%d of the C2 workload x 1000 plies (%d env-steps), 1 thread, %.1f s"
                      % (games2 - 1, steps, dt)}


def load_pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), if present."""
    path = os.path.join(ROOT, "profiles", "r01_env_rollout_pmc.json")
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="games per GPU (C2: 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import tetris_rl_amd as trl
    from tetris_rl_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n = args.games
    eng = trl.Engine(n, ruleset="s2", seed=SEED, device=local_rank)
    eng.reset(parallel.game_streams(rank, n))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        eng.rollout_random(args.warmup)
    barrier()
    t0 = time.perf_counter()
    steps_done = eng.rollout_random(args.steps)          # one launch, K plies of every game; syncs its stream
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = eng.last_kernel_ms()                      # HIP events on the engine's stream

    flags = eng.get_state()[:, 313]
    if (flags & 3).any():
        raise SystemExit("a cap overflowed during the benchmark: results invalid")

    elapsed, steps_total = parallel.reduce_job(elapsed, steps_done, dist, dev)
    kernel_ms, _ = parallel.reduce_job(kernel_ms, 0, dist, dev)

    if rank == 0:
        value = steps_total / elapsed
        per_launch_bytes = BYTES_PER_STEP * (steps_total / world)        # one launch per rank
        achieved = per_launch_bytes / (kernel_ms * 1e-3) / 1e9
        pmc = load_pmc_traffic()
        traffic = None
        if pmc and pmc.get("games") == n and pmc.get("plies") == args.steps:
            traffic = pmc.get("hbm_bytes_per_launch")
        out = {
            "metric": "env-steps/sec (HIP env-step kernel, random policy, no MCTS/net)",
            "value": round(value, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u16",
            "data": "synthetic",
            "config": {"workload": "C2: %d concurrent two-player games per GPU (ruleset s2), uniform-random legal "
                                   "policy, auto-reset on terminal; 1 step = 1 ply of every game" % n,
                       "games_per_gpu": n, "seed": SEED, "plies_timed": args.steps,
                       "env_steps_timed": steps_total, "parallelism": "games sharded by rank, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic,
                         "kernel": "k_rollout", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_step": BYTES_PER_STEP,
                         "note": "integer/latency-bound search; HBM is nominal (SURVEY 8d)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
