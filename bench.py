#!/usr/bin/env python3
"""bench.py -- env-steps/s (BASELINE.json C2) and MCTS moves/s (C3) of the HIP rollout engine.

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

The same run then measures BASELINE.json's C3: 1024 concurrent self-play games per GPU, GPU-resident
MCTS at MAX_ITER=160 (training=True: playout-cap randomisation, root noise, temperature) with the
10-block/16-filter AlphaSame net (random init, torch.manual_seed(0)); reported under "mcts" in the
same JSON line, with the MFMA roofline of the fused residual-tower kernel.

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
PREROLL_PLIES = 300           # input set-up of C2: ~6 game lengths of random play -> stationary mix of game phases


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
    out = {"value": round(steps / dt, 1), "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": "oracle/tetris_oracle.c, games 0..%d of the C2 workload x 1000 plies (%d env-steps), 1 thread, %.1f s"
                     % (games2 - 1, steps, dt)}
    # the same with one game shard per host core: one fresh process per core (the oracle is not
    # re-entrant, and a process that has touched the GPU is not forked) -- `bench.py --cpu-shard C GAMES` below
    try:
        import subprocess
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        cores = min(cores, 16)                                  # a GPU's share of the host (the box shows every core of the node)
        if cores > 1:
            per = max(16, games2 // 3)
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-shard", str(c), str(per)],
                                      stdout=subprocess.PIPE) for c in range(cores)]
            tot = sum(int(p.communicate(timeout=300)[0].decode().split()[0]) for p in procs)
            dt = time.perf_counter() - t0
            out["all_cores"] = {"value": round(tot / dt, 1), "cores": cores,
                                "sample": "%d processes x %d games x 1000 plies, one per core, %.1f s incl. start-up" % (cores, per, dt)}
    except Exception as e:                                      # a baseline, never a reason to lose the bench line
        out["all_cores"] = {"error": repr(e)}
    return out


MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_16x16x4_f32)
MFMA_F16_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: dense BF16 / F16 matrix peak (16 x the fp32 matrix rate)
TRUNK_FLOP_PER_BOARD = 2 * 400 * (25 * 16 + 20 * 144 * 16 + 16)     # conv5x5 + 20 conv3x3 + conv1x1 = 37.2 MFLOP
# k_trunk16s executes every 3x3-conv product as three f16 products (hi hi, lo hi, hi lo) over K padded from 144 to 160
SPLIT_FLOP_PER_BOARD = 2 * 400 * (25 * 16 + 20 * 160 * 16 * 3 + 16)
SPLIT_NOPAD_FLOP_PER_BOARD = 2 * 400 * (25 * 16 + 20 * 144 * 16 * 3 + 16)       # the three products, without the padding of K


def bench_mcts(args, rank, local_rank, world, dist, dev, barrier):
    """C3: moves/s of the GPU search with the AlphaSame(10,16) evaluator"""
    import numpy as np
    import torch
    import tetris_rl_amd as trl
    from tetris_rl_amd import parallel
    n = args.mcts_games
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16))
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16),
                     MAX_ITER=160, training=True)
    # the slots tick in four groups on four streams (selfplay.SelfPlayGroups): the search kernels of some beside the towers of others
    sp = trl.make_self_play(cfg, net, n, seed=SEED, device=local_rank, streams=parallel.game_streams(rank, n),
                            auto_reset=True, use_graph=True, record_capacity=1 << 17)
    groups = len(getattr(sp, "parts", [sp]))
    sp.tick(args.mcts_warmup)
    tk0 = time_tick_kernels(sp, reps=32)                    # the tick's two kernels, live, before ...
    sp.drain()
    st0 = sp.stats()
    barrier()
    t0 = time.perf_counter()
    done = 0
    drained = []
    pending = None
    while done < args.mcts_ticks:
        # records leave the search's buffers inside the timed region and stay in HBM: after every chunk of ticks a drain is
        # queued on each slot group's stream (trl_mcts_drain_records_async: no host round trip), the next chunk right
        # behind it; the host picks the previous chunk's records up on a side stream while the device goes on
        k = min(512, args.mcts_ticks - done)
        sp.tick(k)
        ticket = sp.drain_async()
        if pending is not None:
            drained.append(sp.take(pending)[0])
        pending = ticket
        done += k
    if pending is not None:
        drained.append(sp.take(pending)[0])
    sp.synchronize()
    t_ticks = time.perf_counter() - t0
    # C4: the one exchange step of the path -- every rank's packed (state, pi, z) move records to rank 0: over RCCL
    # GPU -> GPU (dist.gather, no host bounce on the senders), then one copy to rank 0's host
    with torch.cuda.stream(sp.stream):
        mine = torch.cat(drained) if drained else torch.zeros(0, dtype=torch.uint8, device=sp.device)
    sp.synchronize()
    if parallel.active(dist):
        if dev is None:                                # gloo rehearsal: the backend moves host memory
            mine = mine.cpu()
        sizes = parallel.all_gather_ints([mine.numel()], dist, dev)[:, 0]
        parts = parallel.gather_tensor(mine, sizes, dist)
        gathered = sum(int(p.cpu().numel()) for p in parts) if parts is not None else 0
    else:
        gathered = int(mine.cpu().numel())
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("TRL_BENCH_DEBUG"):
        sys.stderr.write("[bench] C3 timed region %.1f ms: ticks + drains %.1f ms, gather + host copy %.1f ms\n"
                         % (elapsed * 1e3, t_ticks * 1e3, (elapsed - t_ticks) * 1e3))
    st1 = sp.stats()
    if st1["overflow"] or st1["records_dropped"] or st1.get("tower_overflow"):
        raise SystemExit("tree / record capacity (or the split towers' activation range) overflowed during the MCTS benchmark: results invalid")
    moves = st1["moves"] - st0["moves"]
    sims = st1["simulations"] - st0["simulations"]
    # live timing of the dominant kernel (fused tower) with events on the stream it runs on
    boards = sp.grids.reshape(-1, 1, 40, 10)               # one tower launch of the tick: both boards of a group's slots
    nb = boards.shape[0]
    with torch.cuda.stream(sp.stream):
        sp.net._trunk_hip(boards, sp.net._folded["packed_tower"])       # the launch alone (trunk() also waits and reads the range guard)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record(sp.stream)
        for _ in range(reps):
            sp.net._trunk_hip(boards, sp.net._folded["packed_tower"])       # the launch alone (trunk() also waits and reads the range guard)
        e1.record(sp.stream)
    e1.synchronize()
    trunk_ms = e0.elapsed_time(e1) / reps
    tk1 = time_tick_kernels(sp, reps=32)                    # ... and after the timed region (the trees are deeper by then)
    tk = None if (tk0 is None or tk1 is None) else (0.5 * (tk0[0] + tk1[0]), 0.5 * (tk0[1] + tk1[1]), tk1[2])
    my_rate = moves / elapsed
    elapsed, moves_total = parallel.reduce_job(elapsed, moves, dist, dev)
    _, sims_total = parallel.reduce_job(elapsed, sims, dist, dev)
    per_rank = None
    if parallel.active(dist):
        per_rank = [round(v / 1e3, 1) for v in parallel.all_gather_ints([int(my_rate * 1e3)], dist, dev)[:, 0]]
    split = getattr(sp.net, "tower_mode", "fp32") == "split"
    algo_tflops = TRUNK_FLOP_PER_BOARD * nb / (trunk_ms * 1e-3) / 1e12
    if split:
        # the tower runs on the f16 matrix instruction with every fp32 operand split in two halves (fp32 accuracy):
        # achieved = f16 matrix FLOPs executed, against the dense f16 peak; the algorithmic (fp32-equivalent) rate beside it
        tflops, peak, kname = SPLIT_FLOP_PER_BOARD * nb / (trunk_ms * 1e-3) / 1e12, MFMA_F16_PEAK_TFLOPS, "k_trunk16s"
    else:
        tflops, peak, kname = algo_tflops, MFMA_F32_PEAK_TFLOPS, "k_trunk16"
    tower_block = {"bound": "mfma", "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                   "frac": round(tflops / peak, 4), "traffic": None, "kernel": kname,
                   "kernel_ms": round(trunk_ms, 4), "flop_per_board": SPLIT_FLOP_PER_BOARD if split else TRUNK_FLOP_PER_BOARD,
                   "boards_per_launch": nb, "algorithmic_tflops": round(algo_tflops, 2),
                   "algorithmic_flop_per_board": TRUNK_FLOP_PER_BOARD,
                   # three f16 products per multiply are part of the method, the padding of K from 144 to 160 is not:
                   "frac_without_k_padding": round(SPLIT_NOPAD_FLOP_PER_BOARD * nb / (trunk_ms * 1e-3) / 1e12 / peak, 4) if split else None,
                   "note": ("f16 matrix instructions on split fp32 operands (a = hi + lo / 2^11; 3 products per multiply, K 144 -> 160): "
                            "the algorithmic rate is %.2f x the fp32 matrix peak of %.1f TFLOP/s that bounded round 1's tower; "
                            "kernel_ms = one launch over ALL boards of a slot group, timed alone after the run -- inside a tick the "
                            "tower skips the boards whose output is cached (about half: tick_kernels.tower_ms)"
                            % (algo_tflops / MFMA_F32_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS)) if split else "fp32 matrix instructions"}
    roofline, second, tick_kernels = tower_block, None, None
    if tk is not None:
        tower_tick_ms, search_ms, slots = tk
        tick_kernels = {"slots_per_launch": slots, "tower_ms": round(tower_tick_ms, 4), "k_expand_select_ms": round(search_ms, 4),
                        "before_timed_region": {"tower_ms": round(tk0[0], 4), "k_expand_select_ms": round(tk0[1], 4)},
                        "after_timed_region": {"tower_ms": round(tk1[0], 4), "k_expand_select_ms": round(tk1[1], 4)},
                        "how": "HIP events on one slot group's stream round its two launches, 32 eager ticks (medians; the first 8 not counted) before and 32 "
                               "after the timed region, the other groups idle, the cost of recording an event (measured in the same "
                               "loop) taken off; the figure is the mean of the two -- what the kernel trace's per-grid average of "
                               "the same run shows (profiles/: kernel_trace_by_grid)"}
        pm = load_pmc("r03_mcts_pmc.json")
        search_bytes = SIM_BYTES * slots
        gbs = search_bytes / (search_ms * 1e-3) / 1e9
        search_block = {"bound": "hbm", "achieved": round(gbs, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 6), "traffic": None, "kernel": "k_expand_select",
                        "kernel_ms": round(search_ms, 4), "simulations_per_launch": slots,
                        "algorithmic_bytes_per_simulation": SIM_BYTES, "survey_bytes_per_simulation": SIM_BYTES_SURVEY,
                        "note": "one workgroup per slot walking a dependent chain (heads, expansion, backup, selection, make_move; "
                                "the leaf's legal-move search rides in the tower launch): latency-bound, neither HBM nor issue "
                                "binds; the HBM fraction is nominal (SURVEY 8d). "
                                "Algorithmic bytes: DESIGN.md 3.3 (the 46 KB policy row of SURVEY's 54 KB is fused away)"}
        if pm and pm.get("kernel") == "k_expand_select":
            scale = slots / float(pm["slots_per_launch"])
            search_block["traffic"] = int((pm["fetch_bytes_per_launch"] + pm["write_bytes_per_launch"]) * scale)
            search_block["counters"] = {k: pm[k] for k in ("wait_any_frac", "l2_hit_rate", "valu_inst_per_wave", "vmem_rd_per_wave",
                                                           "fetch_bytes_per_launch", "write_bytes_per_launch", "slots_per_launch",
                                                           "source") if k in pm}
        # the block named "roofline" is the kernel with the larger share of the tick
        if search_ms >= tower_tick_ms:
            roofline, second = search_block, ("roofline_tower", tower_block)
        else:
            second = ("roofline_search", search_block)
    out = {
        "metric": "MCTS moves/sec at MAX_ITER=160 (GPU search + AlphaSame(10,16) fp32)",
        "value": round(moves_total / elapsed, 1), "unit": "moves/s",
        "simulations_per_sec": round(sims_total / elapsed, 1),
        "ms_per_tick": round(elapsed * 1e3 / args.mcts_ticks, 4),
        "config": {"workload": "C3: %d concurrent self-play games per GPU, MAX_ITER=160, training=True "
                               "(playout cap 400/80, root noise, temperature), AlphaSame blocks=10 filters=16 random init"
                               % n, "games_per_gpu": n, "ticks_timed": args.mcts_ticks, "moves_timed": moves_total,
                   "simulations_timed": sims_total, "slot_groups": groups,
                   # leaves whose mover's board had been evaluated earlier in the same search (the same placement under
                   # another reply of the opponent): their tower output comes from the search's board table, not the tower
                   "leaf_boards_from_the_board_table": round((st1["boards_reused"] - st0["boards_reused"]) / max(sims, 1), 3),
                   "records_to_rank0_mb": round(gathered / 1e6, 1),
                   "parallelism": "games sharded by rank; move records gathered to rank 0 (dist.gather, GPU to GPU) in the timed region"},
        "roofline": roofline,
    }
    if second is not None:
        out[second[0]] = second[1]
    if tick_kernels is not None:
        out["tick_kernels"] = tick_kernels
    if per_rank is not None:
        out["per_rank_value"] = per_rank
    return out


WIDE_FLOP_PER_BOARD = 2 * 400 * (25 * 64 + 40 * 576 * 64 + 64)      # C5 tower: conv5x5 + 40 conv3x3 (64 ch) + conv1x1 = 1.18 GFLOP
WIDE_SPLIT_FLOP_PER_BOARD = 2 * 400 * (25 * 64 + 40 * 576 * 64 * 3 + 64)   # k_trunk_wide_s: three f16 products per multiply


def bench_c5(args, rank, local_rank, world, dist, dev, barrier):
    """C5: the MFMA-bound regime -- AlphaSame(20, 64), MAX_ITER=800, forced playouts + policy-target pruning.
    A move takes 400 or 2000 simulations (playout cap), so a bounded run completes few moves per game: the rate
    reported is simulations/s, with moves/s = simulations/s / 800 (the mean) beside the moves actually completed."""
    import torch
    import tetris_rl_amd as trl
    from tetris_rl_amd import parallel
    n = args.mcts_games
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=20, filters=64)
    net = trl.AlphaSame(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=800, training=True,
                     use_forced_playouts_and_policy_target_pruning=True)
    sp = trl.make_self_play(cfg, net, n, seed=SEED, device=local_rank, streams=parallel.game_streams(rank, n),
                            auto_reset=True, use_graph=True, record_capacity=1 << 16)
    sp.tick(args.c5_warmup)
    sp.drain()
    st0 = sp.stats()
    barrier()
    t0 = time.perf_counter()
    done = 0
    while done < args.c5_ticks:
        k = min(256, args.c5_ticks - done)
        sp.tick(k)
        done += k
        sp.drain()
    sp.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    st1 = sp.stats()
    if st1["overflow"] or st1["records_dropped"] or st1.get("tower_overflow"):
        raise SystemExit("tree / record capacity (or the split towers' activation range) overflowed during the C5 benchmark: results invalid")
    boards = sp.grids.reshape(-1, 1, 40, 10)               # one tower launch of the tick: both boards of a group's slots
    nb = boards.shape[0]
    with torch.cuda.stream(sp.stream):
        sp.net._trunk_hip(boards, sp.net._folded["packed_tower"])       # the launch alone (trunk() also waits and reads the range guard)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record(sp.stream)
        for _ in range(reps):
            sp.net._trunk_hip(boards, sp.net._folded["packed_tower"])       # the launch alone (trunk() also waits and reads the range guard)
        e1.record(sp.stream)
    e1.synchronize()
    trunk_ms = e0.elapsed_time(e1) / reps
    elapsed, moves_total = parallel.reduce_job(elapsed, st1["moves"] - st0["moves"], dist, dev)
    _, sims_total = parallel.reduce_job(elapsed, st1["simulations"] - st0["simulations"], dist, dev)
    split = getattr(sp.net, "tower_mode", "fp32") == "split"
    algo_tflops = WIDE_FLOP_PER_BOARD * nb / (trunk_ms * 1e-3) / 1e12
    if split:
        tflops, peak, kname = WIDE_SPLIT_FLOP_PER_BOARD * nb / (trunk_ms * 1e-3) / 1e12, MFMA_F16_PEAK_TFLOPS, "k_trunk_wide_s<64>"
    else:
        tflops, peak, kname = algo_tflops, MFMA_F32_PEAK_TFLOPS, "k_trunk_wide<64>"
    return {
        "metric": "MCTS simulations/sec at MAX_ITER=800 (GPU search + AlphaSame(20,64) fp32, forced playouts + pruning)",
        "value": round(sims_total / elapsed, 1), "unit": "simulations/s",
        "moves_per_sec_at_mean_800_sims": round(sims_total / elapsed / 800.0, 2),
        "ms_per_tick": round(elapsed * 1e3 / args.c5_ticks, 4),
        "config": {"workload": "C5: %d concurrent self-play games per GPU, MAX_ITER=800, training=True (playout cap "
                               "2000/400), forced playouts + policy-target pruning, AlphaSame blocks=20 filters=64 random init"
                               % n, "games_per_gpu": n, "ticks_timed": args.c5_ticks, "moves_completed": moves_total,
                   "simulations_timed": sims_total, "slot_groups": len(getattr(sp, "parts", [sp])),
                   "leaf_boards_from_the_board_table": round((st1["boards_reused"] - st0["boards_reused"]) /
                                                             max(st1["simulations"] - st0["simulations"], 1), 3)},
        "roofline": {"bound": "mfma", "achieved": round(tflops, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(tflops / peak, 4), "traffic": None, "kernel": kname,
                     "kernel_ms": round(trunk_ms, 4), "flop_per_board": WIDE_SPLIT_FLOP_PER_BOARD if split else WIDE_FLOP_PER_BOARD,
                     "boards_per_launch": nb, "algorithmic_tflops": round(algo_tflops, 2),
                     "algorithmic_flop_per_board": WIDE_FLOP_PER_BOARD,
                     "note": ("f16 matrix instructions on split fp32 operands (3 products per multiply): the algorithmic rate is %.2f x "
                              "the fp32 matrix peak of %.1f TFLOP/s; kernel_ms = one launch over ALL boards of a slot group, timed "
                              "alone after the run -- inside a tick the tower evaluates only the boards that are neither cached "
                              "from the parent position nor in the search's board table"
                              % (algo_tflops / MFMA_F32_PEAK_TFLOPS, MFMA_F32_PEAK_TFLOPS))
                             if split else "fp32 matrix instructions"},
    }


def cpu_baseline_mcts(seconds_target=12.0, blocks=10, filters=16, max_iter=160, forced=False):
    """oracle search (C) + the stub evaluator is not the C3 / C5 workload; the honest CPU statement of it is
    the oracle's search with the same net on PyTorch-CPU.  One game slot, one thread, bounded sample."""
    import ctypes as C
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    import tetris_rl_amd as trl
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=filters)).prepare()
    torch.set_num_threads(1)
    L = op.lib()
    EVAL = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)
    state = np.zeros(320, dtype=np.uint8)
    from tetris_rl_amd import records
    t0 = time.perf_counter()
    evals = [0]

    class Done(Exception):
        pass

    def ev(gptr, vptr, pptr, ctx):
        L.to_export(gptr, state.ctypes.data)
        x = records.game_to_x(state)
        grids = np.stack([x[0].reshape(-1), x[5].reshape(-1)])[:, None, :]
        vec = np.concatenate([x[1].reshape(-1), [x[2], x[3], x[4]], x[6].reshape(-1), [x[7], x[8], x[9]], [x[10]]])[None].astype(np.float32)
        v, p = net.evaluate_packed(torch.from_numpy(grids), torch.from_numpy(vec))
        vptr[0] = float(v[0])
        C.memmove(pptr, p.numpy().ctypes.data, 11583 * 4)
        evals[0] += 1

    cb = EVAL(ev)
    L.to_mcts.argtypes = [C.c_void_p, C.POINTER(op.MctsConfig), EVAL, C.c_void_p, C.POINTER(op.MctsResult)]
    g = op.OracleGame(ruleset=2, seed=SEED, stream=0)
    cfg = op.default_mcts_config(max_iter=max_iter, training=1, use_forced=int(bool(forced)))
    res = op.MctsResult()
    moves = sims = 0
    # whole searches only (a search cannot be cut short through the oracle's interface): at least one, then until the time is up
    while (moves == 0 or time.perf_counter() - t0 < seconds_target) and not g.is_terminal():
        L.to_mcts(g._buf, C.byref(cfg), cb, None, C.byref(res))
        g.make_move(res.move)
        moves += 1
        sims += res.max_iterations
    dt = time.perf_counter() - t0
    return {"value": round(moves / dt, 3), "unit": "moves/s", "cores": 1, "kind": "port",
            "simulations_per_sec": round(sims / dt, 1),
            "sample": "oracle search (C) + same AlphaSame(%d,%d) on PyTorch-CPU 1 thread, game stream 0, %d moves / %d simulations in %.1f s"
                      % (blocks, filters, moves, sims, dt)}


def load_pmc(name="r03_env_rollout_pmc.json"):
    """per-ply figures of k_rollout from the committed rocprofv3 --pmc passes (profiles/): HBM-side bytes per ply and
    game, vector-ALU instructions per wave and ply -- measured at the launch lengths listed in the file"""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


def cpu_shard(c, per):
    """worker of cpu_baseline's all-cores leg: games [c * per, (c + 1) * per) x 1000 plies on the oracle; prints the steps"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle_py as op
    ob = op.OracleBatch(per, ruleset=2, seed=SEED, streams=np.arange(c * per, (c + 1) * per, dtype=np.uint32))
    print(ob.rollout_random(1000))


def _end_process_groups(procs, grace=10.0):
    """terminate exactly the rank processes started here (each leads its own session, so its helpers go with it)"""
    import signal
    import subprocess
    alive = [p for p in procs if p.poll() is None]
    for p in alive:
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError):
            pass
    deadline = time.monotonic() + grace
    for p in alive:
        try:
            p.wait(timeout=max(0.1, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                pass
            p.wait()


def launch_ranks(n, argv, rank_timeout):
    """`python bench.py --gpus N` started as ONE process: this parent makes no GPU call; it starts N fresh rank processes
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run would set them), relays rank 0's JSON line and
    exits non-zero if any rank does -- or if the ranks are not done after `rank_timeout` seconds (a rank parked in a
    collective, the failure a first RCCL run produces): then it says which ranks were still alive, ends them and exits 124.
    The same clean-up runs when this process is told to stop (SIGTERM / SIGINT from a caller's own timeout)."""
    import signal
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    fd, line_path = tempfile.mkstemp(prefix="trl_bench_", suffix=".json")
    os.close(fd)
    procs, rc = [], 0

    def stop(signum, frame):
        sys.stderr.write("bench.py: signal %d: ending the rank processes\n" % signum)
        _end_process_groups(procs)
        os._exit(128 + signum)

    old_handlers = {sg: signal.signal(sg, stop) for sg in (signal.SIGTERM, signal.SIGINT)}
    t_start = time.monotonic()
    try:
        for r in range(n):
            # rank 0 writes the JSON line to TRL_BENCH_JSON; the ranks' stdout (gloo / RCCL banners) goes to stderr, so
            # this process prints the ONE line and nothing else
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TRL_BENCH_JSON=line_path)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=sys.stderr,
                                          start_new_session=True))
        pending = list(procs)
        while pending and not rc:
            for p in list(pending):
                code = p.poll()
                if code is not None:
                    pending.remove(p)
                    rc = rc or code
            if pending and not rc and time.monotonic() - t_start > rank_timeout:
                alive = [procs.index(p) for p in pending]
                sys.stderr.write("bench.py: ranks %s still running after --rank-timeout %.0f s (ranks %s had finished): "
                                 "ending them\n" % (alive, rank_timeout, [r for r in range(n) if r not in alive]))
                rc = 124
            time.sleep(0.05)
    finally:
        _end_process_groups(procs)                          # a failed rank leaves the others inside a collective
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    with open(line_path) as f:
        line = f.read()
    os.unlink(line_path)
    if rc or not line.strip():
        sys.stderr.write("bench.py: a rank exited with code %d\n" % rc)
        raise SystemExit(rc if 0 < rc < 256 else 1)
    sys.stdout.write(line)
    sys.stdout.flush()


def arm_watchdog(seconds, rank):
    """a rank started by torch.distributed.run has no parent of ours to end it: a daemon timer ends a rank that is still
    running after --rank-timeout (a wait inside a collective releases the GIL, so the timer does fire)"""
    import threading

    def fire():
        sys.stderr.write("bench.py: rank %d still running after --rank-timeout %.0f s: exiting\n" % (rank, seconds))
        sys.stderr.flush()
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def bench_c1(args, local_rank):
    """BASELINE config C1, the reference's own case (simulation.py: one self-play game, ruleset s2, AlphaSame(10,16),
    MAX_ITER=160, training=True) through the package's play_game path on ONE slot.  One slot leaves the chip idle; it is the
    configuration the CPU reference is quoted on, so it is reported, not optimised for."""
    import numpy as np
    import torch
    import tetris_rl_amd as trl
    from tetris_rl_amd import selfplay
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=10, filters=16)
    net = trl.AlphaSame(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
    selfplay._run_games(cfg, net, 1, SEED, device=local_rank, streams=np.array([1], dtype=np.uint32))     # warm-up: library, graph capture
    moves = 0
    t0 = time.perf_counter()
    for k in (2, 3):
        games, _ = selfplay._run_games(cfg, net, 1, SEED, device=local_rank, streams=np.array([k], dtype=np.uint32))
        moves += len(games[0]["moves"])
    dt = time.perf_counter() - t0
    return {"metric": "MCTS moves/sec, one self-play game at a time (C1)", "value": round(moves / dt, 2), "unit": "moves/s",
            "config": {"workload": "C1: single self-play game (1 slot), ruleset s2, AlphaSame blocks=10 filters=16, MAX_ITER=160, "
                                   "training=True; 2 games back to back through _run_games (engine set-up and graph capture of each "
                                   "game inside the timed region)", "games": 2, "moves_timed": moves},
            "note": "Python reference on one core of the build container: 0.77 moves/s (SURVEY section 6)"}


def time_tick_kernels(sp, reps=48):
    """HIP events around the two launches of the production tick (tower on the flagged boards, k_expand_select), issued
    eagerly on the slot group's own stream outside the timed region (called once before and once after it: the trees, and
    with them the selection paths, are deeper afterwards) -- the other groups idle, so each kernel has the chip to itself;
    the other groups then make the same number of ticks.  Returns (tower ms, search ms, slots) as medians, or None when
    the tick is not the fused native one."""
    import statistics
    part = getattr(sp, "parts", [sp])[0]
    if not getattr(part, "fused_select", False):
        return None
    torch = part.torch
    for p in getattr(sp, "parts", [sp]):
        p.synchronize()
    # four events per tick: e0 | tower | e1 e1b | search | e2 -- the interval e1 -> e1b holds nothing but the recording of
    # an event (3-4 us), which every other interval contains once as well and which is taken off (without it the shorter
    # kernel read 14 % long against the kernel trace)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]
    with torch.cuda.stream(part.stream):
        for e0, e1, e1b, e2 in ev:
            e0.record(part.stream)
            part._tower()
            e1.record(part.stream)
            e1b.record(part.stream)
            part.engine.mcts_expand_alphasame_select(part.z.data_ptr(), part.vec.data_ptr(), part.heads, part.grids.data_ptr(),
                                                     part.need.data_ptr())
            e2.record(part.stream)
    part.synchronize()
    part.ticks += reps
    over = statistics.median(e1.elapsed_time(e1b) for e0, e1, e1b, e2 in ev[8:])
    tower = statistics.median(e0.elapsed_time(e1) for e0, e1, e1b, e2 in ev[8:]) - over
    search = statistics.median(e1b.elapsed_time(e2) for e0, e1, e1b, e2 in ev[8:]) - over
    for p in getattr(sp, "parts", [sp])[1:]:                # the other groups make the same number of ticks: lock-step rounds
        p.tick(reps)
        p.synchronize()
    return tower, search, part.n


SIM_BYTES = 9500            # algorithmic bytes of one simulation with the policy row fused away
SIM_BYTES_SURVEY = 54000    # the same with the 46 KB softmaxed policy row the reference's data flow reads


def main():
    if len(sys.argv) == 4 and sys.argv[1] == "--cpu-shard":
        return cpu_shard(int(sys.argv[2]), int(sys.argv[3]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--launches", type=int, default=9,
                    help="timed launches of --steps plies each (every one bracketed by barrier + synchronize); the line "
                         "reports the median launch, with the minimum and maximum beside it")
    ap.add_argument("--games", type=int, default=GAMES_PER_GPU, help="games per GPU (C2: 4096)")
    ap.add_argument("--preroll", type=int, default=PREROLL_PLIES,
                    help="plies played while the synthetic input is set up, before the warm-up steps: all games start at ply 0 "
                         "together and random play ends a game after ~50 plies, so the stationary mix of game phases (what "
                         "C2 means by random-queue boards) is reached only after a few game lengths")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mcts", action="store_true", help="skip the C3 (MCTS moves/s) measurement")
    ap.add_argument("--mcts-games", type=int, default=1024, help="concurrent self-play games per GPU (C3: 1024)")
    ap.add_argument("--mcts-ticks", type=int, default=3200, help="timed search ticks (3200 = 20 moves/game at 160 sims)")
    ap.add_argument("--mcts-warmup", type=int, default=200)
    ap.add_argument("--no-c5", action="store_true", help="skip the C5 (20x64 net, MAX_ITER=800) measurement")
    ap.add_argument("--c5-ticks", type=int, default=1200, help="timed search ticks of the C5 measurement")
    ap.add_argument("--c5-warmup", type=int, default=50)
    ap.add_argument("--no-c1", action="store_true", help="skip the C1 (single game) measurement")
    ap.add_argument("--rank-timeout", type=float, default=900.0,
                    help="N > 1: seconds after which ranks that are still running are ended and the run fails (exit 124)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:], args.rank_timeout)   # not under torch.distributed.run: start the ranks here
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        arm_watchdog(args.rank_timeout, rank)
        if str(rank) in os.environ.get("TRL_BENCH_TEST_STALL_RANK", "").split(","):     # test hook: this rank never joins the others
            sys.stderr.write("[bench rank %d] stalling (TRL_BENCH_TEST_STALL_RANK)\n" % rank)
            sys.stderr.flush()
            while True:
                time.sleep(3600)

    # CPU baseline legs first, before this process makes its first GPU call (they start one worker process per core)
    cpu_c2 = cpu_c3 = cpu_c5 = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_c2 = cpu_baseline()
        if not args.no_mcts:
            cpu_c3 = cpu_baseline_mcts()
            if not args.no_c5:
                # a C5 search is 400 or 2000 evaluations of a 2.4 GFLOP net: minutes per move on one core.  The sample runs
                # searches of a tenth of the length (MAX_ITER 80: 40 or 200 simulations) with the same net and switches
                cpu_c5 = cpu_baseline_mcts(seconds_target=10.0, blocks=20, filters=64, max_iter=80, forced=True)
                cpu_c5["sample"] += "; searches at MAX_ITER=80 instead of 800 (the rate is per simulation: one 20x64 evaluation each)"
                cpu_c5 = {"value": cpu_c5["simulations_per_sec"], "unit": "simulations/s", "cores": 1, "kind": "port",
                          "sample": cpu_c5["sample"]}

    import statistics
    import numpy as np
    import torch
    import tetris_rl_amd as trl
    from tetris_rl_amd import parallel

    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    # one rank per GPU; TRL_BENCH_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box:
    # RCCL refuses two ranks on one device)
    backend = os.environ.get("TRL_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit("%d ranks but %d GPUs" % (world, ndev))
    local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else None
    dist_info = {"backend": None, "world_size": 1}
    # TRL_DIST_SINGLE_RANK=1: a process group of this one rank -- the N-rank code path (barriers, reductions and the record
    # gather on the nccl backend with device tensors) rehearsed on a box with one GPU (tests/test_hip_rccl_single.py)
    single_rank_group = world == 1 and os.environ.get("TRL_DIST_SINGLE_RANK", "0") == "1"
    if single_rank_group:
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or single_rank_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stderr.write("[bench rank %d/%d] device_count=%d, bound to cuda:%d (%s), backend=%s, master %s:%s\n"
                         % (rank, world, ndev, local_rank, torch.cuda.get_device_name(local_rank), backend,
                            os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")))
        sys.stderr.flush()
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}     # what the process group saw
        sys.stderr.write("[bench rank %d/%d] process group up: backend %s, world size %d\n"
                         % (rank, world, dist_info["backend"], dist_info["world_size"]))
        sys.stderr.flush()

    n = args.games
    eng = trl.Engine(n, ruleset="s2", seed=SEED, device=local_rank)
    eng.reset(parallel.game_streams(rank, n))
    if args.preroll > 0:
        eng.rollout_random(args.preroll)                     # input set-up: the games reach their stationary phase mix

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        eng.rollout_random(args.warmup)
    # R timed launches of K plies each (1 step = 1 ply of every game; a launch = --steps plies), every launch bracketed
    # by barrier + synchronize on both sides and reduced over the ranks (max time, total steps): the line reports the
    # MEDIAN launch -- one 0.8 ms launch is one sample of a quantity that spreads by a few per cent
    launches = []
    for _ in range(max(1, args.launches)):
        barrier()
        t0 = time.perf_counter()
        steps_done = eng.rollout_random(args.steps)          # one launch, K plies of every game; syncs its stream
        torch.cuda.synchronize()
        barrier()
        wall = time.perf_counter() - t0
        kms = eng.last_kernel_ms()                           # HIP events on the engine's stream
        wall_job, steps_job = parallel.reduce_job(wall, steps_done, dist, dev)
        kms_job, _ = parallel.reduce_job(kms, 0, dist, dev)
        launches.append({"wall": wall_job, "steps": steps_job, "kernel_ms": kms_job, "my_wall": wall, "my_steps": steps_done})
    order = sorted(range(len(launches)), key=lambda i: launches[i]["steps"] / launches[i]["wall"])
    med = launches[order[len(order) // 2]]
    elapsed, steps_total, kernel_ms = med["wall"], med["steps"], med["kernel_ms"]
    rates = [l["steps"] / l["wall"] for l in launches]
    per_rank = None
    if dist is not None:                                      # every rank's own rate of the median launch, for the SCALE check
        pr = parallel.all_gather_ints([int(med["my_steps"]), int(med["my_wall"] * 1e9)], dist, dev)
        per_rank = [round(float(a) / (float(b) * 1e-9), 1) for a, b in pr]

    flags = eng.get_state()[:, 313]
    if (flags & 3).any():
        raise SystemExit("a cap overflowed during the benchmark: results invalid")
    stream_wrap = bool((flags & 4).any())                     # TRL_FLAG_STREAM_WRAP: a slot went through 256 episodes

    eng.close()
    mcts = None if args.no_mcts else bench_mcts(args, rank, local_rank, world, dist, dev, barrier)
    c5 = None if (args.no_mcts or args.no_c5) else bench_c5(args, rank, local_rank, world, dist, dev, barrier)
    c1 = None if (args.no_mcts or args.no_c1 or world > 1) else bench_c1(args, local_rank)

    if rank == 0:
        value = steps_total / elapsed
        per_launch_bytes = BYTES_PER_STEP * (steps_total / world)        # one launch per rank
        achieved = per_launch_bytes / (kernel_ms * 1e-3) / 1e9
        pmc = load_pmc()
        traffic = valu = None
        if pmc and pmc.get("games") == n:
            # HBM-side bytes of one launch: the committed per-ply counter figures scaled to this launch's length
            traffic = int(pmc["hbm_bytes_fixed_per_launch"] + pmc["hbm_bytes_per_ply"] * args.steps)
            # the bound that binds: vector-instruction issue.  every wave64 vector instruction holds its SIMD for
            # VALU_ISSUE_CYCLES cycles (measured, scripts/valu_peak.hip)
            inst = pmc["valu_inst_per_wave_ply"] * pmc.get("waves_per_game", 2) * n * args.steps
            rate = inst / (kernel_ms * 1e-3)
            peak = pmc["simds"] * pmc["clock_ghz"] * 1e9 / pmc["valu_issue_cycles"]
            valu = {"bound": "valu_issue", "achieved": round(rate / 1e9, 2), "peak": round(peak / 1e9, 2),
                    "unit": "G wave-instructions/s", "frac": round(rate / peak, 4),
                    "valu_inst_per_wave_ply": pmc["valu_inst_per_wave_ply"], "source": pmc.get("source")}
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
                       "games_per_gpu": n, "seed": SEED, "plies_timed": args.steps, "preroll_plies": args.preroll,
                       "env_steps_timed": steps_total, "parallelism": "games sharded by rank, no collective"},
            "launches": {"n": len(launches), "plies_per_launch": args.steps, "reported": "median launch",
                         "value_min": round(min(rates), 1), "value_max": round(max(rates), 1),
                         "values": [round(r, 1) for r in rates]},
            "dist": dist_info,
            "roofline": {"bound": "hbm", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "traffic": traffic,
                         "kernel": "k_rollout", "kernel_ms": round(kernel_ms, 4),
                         "algorithmic_bytes_per_step": BYTES_PER_STEP,
                         "note": "integer/latency-bound search; HBM is nominal (SURVEY 8d); the binding bound is roofline_valu"},
        }
        if per_rank is not None:
            out["per_rank_value"] = per_rank
        if stream_wrap:
            out["config"]["stream_wrap"] = "a slot passed 256 episodes and replays its first games (TRL_FLAG_STREAM_WRAP)"
        if valu is not None:
            out["roofline_valu"] = valu
        if mcts is not None:
            out["mcts"] = mcts
        if c5 is not None:
            out["mcts_c5"] = c5
            if cpu_c5 is not None:
                out["mcts_c5"]["cpu_baseline"] = cpu_c5
        if c1 is not None:
            out["c1"] = c1
        if cpu_c2 is not None:
            out["cpu_baseline"] = cpu_c2
            if mcts is not None and cpu_c3 is not None:
                out["mcts"]["cpu_baseline"] = cpu_c3
        if os.environ.get("TRL_BENCH_JSON"):                  # started by launch_ranks: the parent prints the line
            with open(os.environ["TRL_BENCH_JSON"], "w") as f:
                f.write(json.dumps(out) + "\n")
        else:
            print(json.dumps(out), flush=True)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
