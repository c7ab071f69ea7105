"""Self-play data generation behind the reference's Config / play_game / make_training_set /
self_play_loop interface (ai.py:62-216, 1570-1699, 1809-1869, 2117-2198; seams S1-S4, S9 of
SURVEY.md section 8b).

All games of a set run concurrently as slots of one Engine; every tick is
    trl_mcts_select -> AlphaSame.evaluate_packed (PyTorch-ROCm) -> trl_mcts_expand
on one HIP stream, captured once into a HIP graph and replayed.  Records come back from the
device as (position, root children (move, N)) and are turned into the reference's 13-field
lists on the host (records.py).
"""
import json
import math
import os
from pathlib import Path

import numpy as np

from . import engine as _eng
from . import nets, records


# ---- S1: Config ---------------------------------------------------------------------------------------
_CONFIG_DEFAULTS = dict(           # ai.py:63-138, same names, same defaults
    visual=True, model_version=7.0, data_version=2.9, ruleset="s2", model="keras", use_tflite=True,
    tflite_num_threads=2, batched_inference=False, model_config=None, move_algorithm="convolutional",
    use_tanh=False, training_games=100, training_loops=1, sets_to_train_with=10, battle_games=200,
    gating_threshold=0.52, gating_threshold_type="moreorequal", MAX_ITER=400, CPUCT=0.75, DPUCT=1,
    FpuStrategy="reduction", FpuValue=0.1, use_root_softmax=True, RootSoftmaxTemp=1.1, temperature=0.1,
    training=False, learning_rate=0.001, weight_decay=0.0, epochs=1, batch_size=64, data_loading_style="merge",
    decay_factor=0.9, augment_data=True, shuffle=True, use_experimental_features=False, save_all=False,
    loss_weights=[1, 1], use_random_starting_moves=False, use_playout_cap_randomization=True,
    playout_cap_chance=0.25, playout_cap_mult=5, use_dirichlet_noise=True, DIRICHLET_ALPHA=0.1, DIRICHLET_S=25,
    DIRICHLET_EXPLORATION=0.25, use_dirichlet_s=True, use_forced_playouts_and_policy_target_pruning=False,
    CForcedPlayout=1,
)


def storage_root():
    """ai.py:57-59: Path.cwd().parent / 'Storage' (overridable with TRL_STORAGE)"""
    return Path(os.environ.get("TRL_STORAGE", Path.cwd().parent / "Storage"))


class Config:
    """Same keyword arguments and derived properties as the reference's Config; an unknown
    keyword raises TypeError as a Python signature would."""

    def __init__(self, **kwargs):
        for k in kwargs:
            if k not in _CONFIG_DEFAULTS:
                raise TypeError("Config.__init__() got an unexpected keyword argument %r" % k)
        for k, v in _CONFIG_DEFAULTS.items():
            setattr(self, k, kwargs.get(k, list(v) if isinstance(v, list) else v))
        if self.model_config is None:
            self.model_config = nets.AuxBaseResNetConfig()          # the reference's default (ai.py:83)

    def copy(self):
        return Config(**{k: getattr(self, k) for k in _CONFIG_DEFAULTS})

    @property
    def model_dir(self):
        sub = "pytorch_models" if self.model == "pytorch" else "models"
        return "%s/%s/%s.%s" % (storage_root(), sub, self.ruleset, self.model_version)

    @property
    def data_dir(self):
        return "%s/data/%s.%s" % (storage_root(), self.ruleset, self.data_version)

    @property
    def value_max(self):
        return 1

    @property
    def value_mid(self):
        return 0 if self.use_tanh else 0.5

    @property
    def value_min(self):
        return -1 if self.use_tanh else 0

    def negate_value(self, value):
        return -value if self.use_tanh else 1 - value


def search_options(config, **extra):
    """Config -> trl_mcts_config"""
    if config.move_algorithm not in _eng.MOVE_ALGORITHMS:      # the reference raises the same for a name it does not know
        raise ValueError("Unknown algorithm: %s (the lossy generators 'faster-but-loss' / 'harddrop' are not built)"
                         % config.move_algorithm)
    if config.FpuStrategy not in ("reduction", "absolute"):
        raise ValueError("Unknown FpuStrategy: %s" % config.FpuStrategy)
    opts = dict(max_iter=int(config.MAX_ITER), training=int(bool(config.training)),
                fpu_reduction=int(config.FpuStrategy == "reduction"), use_root_softmax=int(bool(config.use_root_softmax)),
                use_playout_cap=int(bool(config.use_playout_cap_randomization)),
                use_dirichlet=int(bool(config.use_dirichlet_noise)), use_dir_s=int(bool(config.use_dirichlet_s)),
                use_forced=int(bool(config.use_forced_playouts_and_policy_target_pruning)),
                save_all=int(bool(config.save_all)), use_tanh=int(bool(config.use_tanh)),
                random_start=int(bool(config.use_random_starting_moves)),
                move_algorithm=_eng.MOVE_ALGORITHMS[config.move_algorithm], cpuct=float(config.CPUCT), dpuct=float(config.DPUCT),
                fpu_value=float(config.FpuValue), root_softmax_temp=float(config.RootSoftmaxTemp),
                temperature=float(config.temperature), playout_cap_chance=float(config.playout_cap_chance),
                playout_cap_mult=float(config.playout_cap_mult), dir_alpha=float(config.DIRICHLET_ALPHA),
                dir_s=float(config.DIRICHLET_S), dir_expl=float(config.DIRICHLET_EXPLORATION),
                c_forced=float(config.CForcedPlayout))
    opts.update(extra)
    return _eng.mcts_config(**opts)


def fresh_seed():
    """The reference plays on CPython's global generator, seeded from the OS: every set and every match sees fresh
    randomness.  Here a run is a function of (seed, stream), so a seed that is not given is drawn from the OS."""
    return int.from_bytes(os.urandom(8), "little") >> 1


def derive_seed(base, *parts):
    """a reproducible seed per (base seed, set number, model number, purpose): a resumed loop does not replay the games
    of an earlier set, and successive gating matches do not share their piece / garbage streams"""
    import hashlib
    h = hashlib.blake2b(repr((int(base),) + tuple(parts)).encode(), digest_size=8).digest()
    return int.from_bytes(h, "little") >> 1


# ---- the tick loop ----------------------------------------------------------------------------------------
FUSE_SELECT = os.environ.get("TRL_FUSE_SELECT", "1") != "0"   # native tick: expand + next select as one launch
DRAIN_HEAD_WORDS = 9            # trl_mcts_drain_records_async's dev_head: [0] records, [1..8] search counters at that drain
DRAIN_HEAD_FIELDS = ("simulations", "moves", "games", "records_dropped", "max_nodes_used", "overflow", "ticks", "boards_reused")
LEGAL_IN_TOWER = os.environ.get("TRL_LEGAL_IN_TOWER", "1") != "0"   # the fused tick's legal-move search in its tower launch
GRAPH_TICKS = int(os.environ.get("TRL_GRAPH_TICKS", "8"))      # ticks per launch of the long tick graph (1: single-tick graph only)


class SelfPlay:
    """n_slots concurrent self-play games: Engine + net + HIP-graph tick."""

    def __init__(self, config, network, n_slots, seed=0, device=0, streams=None, auto_reset=True, use_graph=True,
                 max_nodes=0, record_capacity=0, policy_head="legal", tower_cache=True, device_heads=True, episodes=(0, 0),
                 fork_legal=None, hip_stream=None):
        """policy_head: "legal" = only the legal moves' logits are formed, inside the expand kernel
        (trl_mcts_expand_head); "full" = the reference's data flow, softmax over all 11 583 logits."""
        if policy_head not in ("legal", "full"):
            raise ValueError("policy_head must be 'legal' or 'full'")
        self.policy_head = policy_head
        self.tower_cache = bool(tower_cache)     # reuse the unchanged board's tower output (trl_mcts_zcache); exact
        import torch
        self.torch = torch
        self.config = config
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.net = network.to(self.device).prepare()
        self.n = int(n_slots)
        self.stream = hip_stream if hip_stream is not None else torch.cuda.Stream(device=self.device)
        self.engine = _eng.Engine(self.n, ruleset=config.ruleset, seed=seed, device=device, stream=self.stream.cuda_stream)
        self.engine.reset(streams)
        # episodes = (base, extra): slot s plays base + (s < extra) games, then idles (0, 0: no limit)
        self.engine.mcts_create(search_options(config, auto_reset=int(bool(auto_reset)), max_nodes=max_nodes,
                                               record_capacity=record_capacity, episodes_base=int(episodes[0]),
                                               episodes_extra=int(episodes[1])))
        self._rec_dev = None
        self._acc = None
        with torch.cuda.stream(self.stream):
            self.grids = torch.zeros((2, self.n, _eng.GRID_FLOATS), dtype=torch.float32, device=self.device)
            self.vec = torch.zeros((self.n, _eng.VEC_FLOATS), dtype=torch.float32, device=self.device)
            self.need = torch.ones(2 * self.n, dtype=torch.uint8, device=self.device)
        self.tower_cache = self.tower_cache and policy_head == "legal" and getattr(self.net, "supports_tower_cache", False)
        if self.tower_cache:
            self.engine.mcts_zcache_dim(self.net.tower_floats_per_board)
        # every head of AlphaSame inside the expand kernel (trl_mcts_expand_alphasame): the tick is three HIP kernels
        self.heads = self.net.device_heads() if (self.tower_cache and device_heads and hasattr(self.net, "device_heads")) else None
        if self.heads is not None:
            with torch.cuda.stream(self.stream):
                self.z = torch.zeros((2 * self.n, _eng.GRID_FLOATS), dtype=torch.float32, device=self.device)
        # the leaf's legal-move search beside the tower (trl_mcts_fork_legal), opt-in (TRL_FORK_LEGAL=1): bit-identical, but
        # measured no faster -- 0.249 against 0.242 ms per C3 tick: the search waves take from the tower what they save
        if fork_legal is None:
            fork_legal = os.environ.get("TRL_FORK_LEGAL", "0") != "0"
        self.fork_legal = bool(fork_legal) and self.tower_cache
        if self.fork_legal:
            self.engine.mcts_fork_legal(True)
        self.stream.synchronize()
        self.graph = None
        self.graph_many = None
        self.native_graph = False
        self.fused_select = False
        self.legal_in_tower = False
        self.ticks = 0
        if use_graph:
            self._capture()

    def _tick_body(self):
        e = self.engine
        # v / p / x / z live in the stream-ordered allocator (a fixed address inside a captured graph)
        if self.fused_select:
            # the selection of this simulation was made by the previous tick's launch (or by _capture for the first one)
            self._tower()
            e.mcts_expand_alphasame_select(self.z.data_ptr(), self.vec.data_ptr(), self.heads, self.grids.data_ptr(), self.need.data_ptr())
            return
        if self.tower_cache:
            e.mcts_select_need(self.grids.data_ptr(), self.vec.data_ptr(), self.need.data_ptr())
            if self.heads is not None:
                # all-native tick (three kernels of the library, no PyTorch op): select -> tower -> expand;
                # the rows of self.z that the tower skips are never read (the expand kernel takes them from its cache)
                self.net.tower_need(self.grids, self.need, out=self.z)
                e.mcts_expand_alphasame(self.z.data_ptr(), self.vec.data_ptr(), self.heads)
                return
            z = self.net.tower_need(self.grids, self.need)
            e.mcts_zcache(z.data_ptr())
            v, x = self.net.evaluate_head_inputs(self.grids, self.vec, z)
            ph = self.net.policy_head
            e.mcts_expand_head(v.data_ptr(), x.data_ptr(), x.shape[1], ph.weight.data_ptr(), ph.bias.data_ptr())
            return
        e.mcts_select(self.grids.data_ptr(), self.vec.data_ptr())
        if self.policy_head == "full":
            v, p = self.net.evaluate_packed(self.grids, self.vec)
            e.mcts_expand(v.data_ptr(), p.data_ptr())
        else:
            v, x = self.net.evaluate_head_inputs(self.grids, self.vec)
            ph = self.net.policy_head
            e.mcts_expand_head(v.data_ptr(), x.data_ptr(), x.shape[1], ph.weight.data_ptr(), ph.bias.data_ptr())

    def _tower(self):
        """the tower launch of the fused tick: the boards the pending selection flagged -- and, with legal_in_tower, the
        legal-move search of its leaves in the same launch"""
        if self.legal_in_tower:
            self.net.tower_legal(self.engine, self.grids, self.need, out=self.z)
        else:
            self.net.tower_need(self.grids, self.need, out=self.z)

    def _capture(self):
        torch = self.torch
        if self.tower_cache and self.heads is not None:
            # the tick is library kernels only: capture it with the library's own HIP graph (trl_graph_*); a
            # torch.cuda.CUDAGraph replay adds half a dozen small generator-state kernels of its own per replay
            with torch.cuda.stream(self.stream):
                if FUSE_SELECT and not self.fork_legal and self.config.move_algorithm == "convolutional":
                    # expand of one simulation and select of the next in one launch (trl_mcts_expand_alphasame_select):
                    # the first selection is made here, every tick after it is {tower, expand + select}
                    if LEGAL_IN_TOWER and getattr(self.net, "supports_tower_legal", False) and self.grids.is_contiguous():
                        # the leaf's legal-move search rides in the tower launch (trl_mcts_legal_in_tower): it needs the
                        # leaf's position only, and its slow cases then overlap the tower instead of ending the search launch
                        self.engine.mcts_legal_in_tower(True)
                        self.legal_in_tower = True
                    self.engine.mcts_select_need(self.grids.data_ptr(), self.vec.data_ptr(), self.need.data_ptr())
                    self.fused_select, self._select_epoch = True, self.engine.state_epoch
                self.engine.graph_begin()
                self._tick_body()
                self.engine.graph_end()
                if GRAPH_TICKS > 1:                   # the same tick GRAPH_TICKS times in one graph: no idle stream between them
                    self.engine.graph_begin()
                    for _ in range(GRAPH_TICKS):
                        self._tick_body()
                    self.engine.graph_end_many(GRAPH_TICKS)
            self.native_graph = True
            return
        with torch.cuda.stream(self.stream):
            # one eager evaluation outside the capture warms up MIOpen / rocBLAS handles and workspaces
            self.net.evaluate_packed(self.grids, self.vec)
        self.stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            self._tick_body()
        self.graph = g
        if GRAPH_TICKS > 1:                           # and GRAPH_TICKS ticks in one graph: no idle stream between them
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self.stream):
                for _ in range(GRAPH_TICKS):
                    self._tick_body()
            self.graph_many = g

    def tick(self, n=1):
        torch = self.torch
        if self.fused_select and self._select_epoch != self.engine.state_epoch:
            # the game states were replaced from outside (set_state / reset / step) after the pending selection was made:
            # drop it, start every slot's search over from the new states and select again
            with torch.cuda.stream(self.stream):
                self.engine.mcts_restart()
                self.engine.mcts_select_need(self.grids.data_ptr(), self.vec.data_ptr(), self.need.data_ptr())
            self._select_epoch = self.engine.state_epoch
        if self.native_graph:
            self.engine.graph_launch(n)
            self.ticks += n
            return
        with torch.cuda.stream(self.stream):           # graph replays launch on the current stream
            left = n
            if self.graph_many is not None:
                for _ in range(left // GRAPH_TICKS):
                    self.graph_many.replay()
                left %= GRAPH_TICKS
            for _ in range(left):
                if self.graph is not None:
                    self.graph.replay()
                else:
                    self._tick_body()
        self.ticks += n

    def synchronize(self):
        self.stream.synchronize()

    def drain(self):
        self.stream.synchronize()
        return self.engine.mcts_drain_records()

    def drain_dev(self, then_tick=0):
        """pending records as a uint8 tensor in HBM (n * 2392 bytes): the payload of the RCCL gather to rank 0.
        then_tick: ticks to enqueue right after the copy, before this call waits for it -- the device goes on searching"""
        torch = self.torch
        cap = self.engine._rec_cap
        with torch.cuda.stream(self.stream):
            if self._rec_dev is None:
                self._rec_dev = torch.empty(cap * _eng.RECORD_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
            n = self.engine.mcts_drain_records_dev(self._rec_dev.data_ptr(), cap)
            out = self._rec_dev[:n * _eng.RECORD_DTYPE.itemsize].clone()
            done = torch.cuda.Event()
            done.record(self.stream)
        if then_tick > 0:
            self.tick(then_tick)
        # the clone was enqueued on this object's stream; its readers (torch.cat, the RCCL gather) run on theirs
        done.synchronize()
        return out

    def drain_async(self, slot_offset=0):
        """Queue "pending records -> one of two accumulators in HBM" behind the ticks queued so far and return a ticket;
        nothing is waited for, so the next chunk of ticks can be queued right behind it (trl_mcts_drain_records_async).
        take(ticket) hands the records over.  Two accumulators: a ticket must be taken before the next but one is drawn."""
        torch = self.torch
        cap = self.engine._rec_cap
        if self._acc is None:
            with torch.cuda.stream(self.stream):
                self._acc = [(torch.empty(cap * _eng.RECORD_DTYPE.itemsize, dtype=torch.uint8, device=self.device),
                              torch.zeros(DRAIN_HEAD_WORDS, dtype=torch.int64, device=self.device)) for _ in range(2)]
            self._acc_busy = [False, False]
            self._acc_next = 0
            self._side = torch.cuda.Stream(device=self.device)
        i = self._acc_next
        if self._acc_busy[i]:
            raise RuntimeError("drain_async: take() the ticket drawn two drains ago first")
        buf, head = self._acc[i]
        self.engine.mcts_drain_records_async(buf.data_ptr(), cap, head.data_ptr(), slot_offset)
        done = torch.cuda.Event()
        done.record(self.stream)
        self._acc_busy[i] = True
        self._acc_next = i ^ 1
        return i, done

    def take(self, ticket):
        """(records of a drain_async ticket as a uint8 tensor in HBM, the search counters as of that drain).  Waits for
        that drain only -- on a side stream, so neither the wait nor the copies queue behind ticks enqueued since."""
        torch = self.torch
        i, done = ticket
        buf, head = self._acc[i]
        with torch.cuda.stream(self._side):
            self._side.wait_event(done)
            h = head.cpu()                                   # (synchronises the side stream)
            n = int(h[0])
            out = buf[:n * _eng.RECORD_DTYPE.itemsize].clone()
            head.zero_()
        self._side.synchronize()
        # the clone lives in the side stream's pool and is read by the caller's stream (torch.cat, the copy in front of the
        # RCCL gather): the allocator must not hand the block out again before that stream has used it
        out.record_stream(torch.cuda.current_stream(self.device))
        self._acc_busy[i] = False
        return out, {k: int(h[1 + j]) for j, k in enumerate(DRAIN_HEAD_FIELDS)}

    def stats(self):
        """trl_mcts_stats + "tower_overflow": boards (since the last call, on this device) whose activations left the f16
        range of the split tower kernels -- the searches that evaluated them ran on wrong priors; callers treat it like a
        tree overflow (the remedy: network.tower_mode = 'fp32')"""
        self.stream.synchronize()
        st = self.engine.mcts_stats()
        st["tower_overflow"] = _eng.tower_overflow(self.device.index or 0, reset=True) if getattr(self.net, "tower_mode", "") == "split" else 0
        return st


def concurrent_streams(torch, device, count, spin_ms=0.4, tries=24):
    """`count` HIP streams of `device` on which kernels really run side by side.  The runtime maps streams onto a few
    hardware queues (4 by default) by an internal least-used rule; two streams that land on one queue execute their
    kernels one after the other, and slot groups ticking on such a pair lose everything the grouping is for (C3 with four
    groups: 0.25 ms per round instead of 0.16, about one process in three).  So: take streams from torch's pool one by one
    and keep a candidate only if a spin kernel on it overlaps a spin kernel on every stream kept so far."""
    import time
    cycles = int(spin_ms * 2.0e6)                       # ~2 GHz: a spin of spin_ms
    kept, rejected = [], []

    def overlaps(a, b):
        for s in (a, b):
            s.synchronize()
        t0 = time.perf_counter()
        for s in (a, b):
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        for s in (a, b):
            s.synchronize()
        both = time.perf_counter() - t0
        t0 = time.perf_counter()
        with torch.cuda.stream(a):
            torch.cuda._sleep(cycles)
        a.synchronize()
        one = time.perf_counter() - t0
        return both < 1.5 * one

    with torch.cuda.device(device):
        torch.cuda._sleep(1000)                         # the first call compiles / loads the spin kernel
        torch.cuda.synchronize()
        while len(kept) < count and len(kept) + len(rejected) < tries:
            c = torch.cuda.Stream(device=device)
            (kept if all(overlaps(c, k) for k in kept) else rejected).append(c)
        while len(kept) < count:                        # no more luck: better a shared queue than no stream
            kept.append(rejected.pop() if rejected else torch.cuda.Stream(device=device))
    return kept


class SelfPlayGroups:
    """The slots of a job as G independently ticking SelfPlay groups, each with its own engine, stream and tick graphs:
    the latency-bound search kernels of one group run beside the throughput-bound tower of another (C3, 1024 slots:
    0.201 -> 0.178 ms per round of ticks with two groups, scripts/mcts_groups_probe.py).  That needs the search kernel to
    fit the registers the tower's two workgroups per CU leave free (112 VGPRs: k_expand_select).  Same interface as
    SelfPlay for the callers that run whole jobs (tick / drain / drain_dev / stats / synchronize); group g owns the global
    slots [g * m, (g + 1) * m) with their RNG streams, so the games -- and, after the slot numbers are shifted back, the
    records -- are those of one SelfPlay over all slots."""

    def __init__(self, config, network, n_slots, groups, streams=None, episodes=(0, 0), **kw):
        if n_slots % groups:
            raise ValueError("n_slots must be a multiple of the number of groups")
        self.n, self.m = int(n_slots), int(n_slots) // groups
        if streams is None:
            streams = np.arange(self.n, dtype=np.uint32)
        streams = np.ascontiguousarray(streams, dtype=np.uint32)
        base, extra = int(episodes[0]), int(episodes[1])
        import torch
        hip_streams = concurrent_streams(torch, torch.device("cuda", kw.get("device", 0)), groups)
        self.parts = []
        for g in range(groups):
            lo = g * self.m
            part_extra = min(max(extra - lo, 0), self.m)           # slot s plays base + (s < extra) games, s global
            self.parts.append(SelfPlay(config, network, self.m, streams=streams[lo:lo + self.m], episodes=(base, part_extra),
                                       hip_stream=hip_streams[g], **kw))
        p0 = self.parts[0]
        self.torch, self.device, self.net, self.config = p0.torch, p0.device, p0.net, config
        self.stream, self.grids, self.heads = p0.stream, p0.grids, p0.heads        # group 0's: what a probe of the tower looks at
        self.ticks = 0

    def tick(self, n=1):
        for p in self.parts:
            p.tick(n)
        self.ticks += n

    def synchronize(self):
        for p in self.parts:
            p.synchronize()

    def drain(self):
        out = []
        for g, p in enumerate(self.parts):
            r = p.drain()
            if len(r):
                r = r.copy()
                r["slot"] += g * self.m
                out.append(r)
        return np.concatenate(out) if out else np.zeros(0, dtype=_eng.RECORD_DTYPE)

    def drain_dev(self, then_tick=0):
        """group by group: a group that has handed over its records goes on ticking (then_tick) while the next is drained"""
        torch, size = self.torch, _eng.RECORD_DTYPE.itemsize
        out = []
        for g, p in enumerate(self.parts):
            t = p.drain_dev(then_tick)
            if t.numel() and g:
                t.view(-1, size)[:, :4].view(torch.int32).add_(g * self.m)            # the record's slot field (first word)
            out.append(t)
        self.ticks += then_tick
        res = torch.cat(out) if len(out) > 1 else out[0]                              # (current stream: the copies are complete)
        torch.cuda.current_stream(self.device).synchronize()
        return res

    def drain_async(self, slot_offset=0):
        """every group's SelfPlay.drain_async (the records carry global slot numbers when they arrive); one ticket"""
        return [p.drain_async(slot_offset + g * self.m) for g, p in enumerate(self.parts)]

    def take(self, ticket):
        got = [p.take(t) for p, t in zip(self.parts, ticket)]
        torch = self.torch
        recs = torch.cat([r for r, _ in got]) if len(got) > 1 else got[0][0]
        torch.cuda.current_stream(self.device).synchronize()
        sts = [st for _, st in got]
        out = {k: sum(st[k] for st in sts) for k in sts[0]}
        out["ticks"] = sts[0]["ticks"]
        out["max_nodes_used"] = max(st["max_nodes_used"] for st in sts)
        return recs, out

    def stats(self):
        sts = [p.stats() for p in self.parts]
        out = {k: sum(st[k] for st in sts) for k in sts[0]}
        out["ticks"] = sts[0]["ticks"]                                   # rounds of ticks, not their sum over the groups
        out["max_nodes_used"] = max(st["max_nodes_used"] for st in sts)
        return out


def slot_groups(config, network, n_slots):
    """how many groups a job's slots are ticked in: TRL_SLOT_GROUPS, or where it pays -- the all-native tick of a
    16-filter AlphaSame (two tower workgroups per CU leave registers for the search kernel): 4 groups from 1024 slots
    (C3: 0.160 ms per round of ticks against 0.177 with 2 and 0.20 with 1), 2 from 512; the 64-filter AlphaSame: 2 from 1024"""
    env = os.environ.get("TRL_SLOT_GROUPS", "auto")
    if env != "auto":
        return max(1, int(env))
    mc = getattr(network, "model_config", None)
    native16 = type(network).__name__ == "AlphaSame" and getattr(mc, "filters", 0) == 16 and getattr(mc, "kernels", 1) == 1
    if native16 and n_slots >= 1024 and n_slots % 4 == 0:
        return 4
    # the 64-filter tower owns its CU (138 KB of LDS), but since the search's board table leaves it only a third of the
    # leaves' boards, a launch over one group's live boards ends in a partly filled round of workgroups -- which the other
    # group's launch fills (C5, 1024 slots: 0.81 -> 1.03 M simulations/s with two groups, 1.00 with four)
    native64 = type(network).__name__ == "AlphaSame" and getattr(mc, "filters", 0) == 64 and getattr(mc, "kernels", 1) == 1
    if native64 and n_slots >= 1024 and n_slots % 2 == 0 and getattr(network, "tower_mode", "") == "split":
        return 2
    return 2 if (native16 and n_slots >= 512 and n_slots % 2 == 0) else 1


def make_self_play(config, network, n_slots, **kw):
    """SelfPlay, or SelfPlayGroups over the same slots when slot_groups() says so"""
    g = slot_groups(config, network, n_slots)
    return SelfPlayGroups(config, network, n_slots, g, **kw) if g > 1 else SelfPlay(config, network, n_slots, **kw)


class GameCollector:
    """Groups the device's record stream into finished games and builds the reference's per-game
    `data` lists (ai.py:1613-1695)."""

    def __init__(self, config, as_text=False, packed_only=False):
        self.config = config
        self.as_text = as_text    # build each record as JSON text (records.record_texts_for_move) instead of lists
        self.packed_only = packed_only   # keep only the packed saved moves (no reference-format records at all)
        self.open = {}            # (slot, episode) -> list of move records
        self.finished = []        # dicts: key, data | texts, stats, plies, winner

    def add(self, recs):
        for r in recs:
            key = (int(r["slot"]), int(r["episode"]))
            if r["kind"] == _eng.REC_MOVE:
                self.open.setdefault(key, []).append(r.copy())
            else:
                moves = sorted(self.open.pop(key, []), key=lambda x: int(x["ply"]))
                game_data, saved = [[], []], [[], []]
                per_move = records.record_texts_for_move if self.as_text else records.records_for_move
                for m in moves:
                    if m["saved"]:
                        n = int(m["n_children"])
                        if not self.packed_only:
                            game_data[int(m["state"][312])] += per_move(m["state"], m["moves"][:n], m["visits"][:n],
                                                                        augment=self.config.augment_data)
                        saved[int(m["state"][312])].append((m["state"].copy(), m["moves"][:n].copy(), m["visits"][:n].copy()))
                c = self.config
                finish = records.finish_game_texts if self.as_text else records.finish_game
                data = finish(game_data, int(r["winner"]), c.value_min, c.value_mid, c.value_max)
                p0 = records.unpack_player(r["state"][0:144])
                stats = {"app": p0["lines_sent"] / max(p0["pieces"], 1), "dspp": p0["lines_cleared"] / max(p0["pieces"], 1)}
                w = int(r["winner"])
                packed = [(st, mv, vis, c.value_mid if w == -1 else (c.value_max if w == pl else c.value_min))
                          for pl in range(2) for st, mv, vis in saved[pl]]          # the set file's record order
                self.finished.append(dict(key=key, data=data, packed=packed, stats=stats, plies=int(r["ply"]), winner=int(r["winner"]),
                                          moves=[int(m["chosen"]) for m in moves]))


# ---- model / data numbering (ai.py:2306-2355) ------------------------------------------------------------------
def _highest_number(path, strip=None):
    best = -1
    if not os.path.isdir(path):              # nothing stored yet; reading a number must not need a writable Storage/
        return best
    for name in os.listdir(path):
        stem = name[:-len(strip)] if (strip and name.endswith(strip)) else name.split(".")[0]
        if stem.isdigit():
            best = max(best, int(stem))
    return best


def highest_model_number(config):
    return _highest_number(config.model_dir, ".pt" if config.model == "pytorch" else None)


def highest_data_number(config):
    """newest set number: {data_dir}/{n}.txt as in the reference, or {data_dir}_packed/{n}.npz of a set written with
    write_json=False"""
    return max(_highest_number(config.data_dir), _highest_number(config.data_dir.rstrip("/") + "_packed"))


def instantiate_network(config, nn_generator=None, show_summary=False, save_network=True):
    """ai.py:1033-1085, PyTorch branch: random-init net of the class the model_config names, saved as {n}.pt"""
    import torch
    net = nn_generator(config.model_config) if nn_generator else nets.model_for_config(config.model_config)
    if save_network:
        os.makedirs(config.model_dir, exist_ok=True)
        torch.save(net.state_dict(), "%s/%d.pt" % (config.model_dir, highest_model_number(config) + 1))
    return net


def load_best_model(config):
    n = highest_model_number(config)
    if n < 0:
        return instantiate_network(config)
    return nets.load_reference_checkpoint("%s/%d.pt" % (config.model_dir, n), config.model_config)


# ---- S3 / S4: make_training_set, play_game --------------------------------------------------------------------------
def _chunk_ticks(config, slots):
    """ticks between two drains: each drain is a stream sync + copies, ~2 ms against 0.4 ms per tick at 1024 slots; the
    device record buffer (default 16 384) must hold what finishes in between: <= slots * chunk / shortest search + games"""
    shortest = config.MAX_ITER
    if config.training and config.use_playout_cap_randomization:
        shortest = max(1, int(config.MAX_ITER / (config.playout_cap_chance * (config.playout_cap_mult - 1) + 1)))
    if config.use_random_starting_moves:
        shortest = 1
    return max(16, min(256, (16384 * shortest) // (2 * slots)))


def _play_collect(config, network, num_games, seed, device=0, max_slots=1024, streams=None, as_text=False, packed_only=False,
                  dist=None):
    """Plays this process's `num_games` games and returns them as finished games (GameCollector) in (episode, slot) order.
    Slot s plays a fixed quota of num_games // slots (+1 for the first num_games % slots slots) consecutive games --
    which games make up a set does not depend on which happen to finish first.
    The host's record conversion overlaps the device: the next ticks are enqueued before the records of the previous
    chunk are turned into training records.
    With an initialised torch.distributed group (`dist`) every rank runs this loop on its own games; after every chunk
    the ranks' new records are gathered to rank 0 -- over RCCL straight from HBM (SelfPlay.drain_dev, no host bounce on
    the senders), over gloo from host memory -- while the next chunk is already running, and rank 0 converts them.
    Rank 0 returns the games of ALL ranks (slot = rank * slots + local slot), the others []."""
    from . import parallel
    multi = parallel.active(dist)
    rank, world = (dist.get_rank(), dist.get_world_size()) if multi else (0, 1)
    slots = min(int(num_games), int(max_slots))
    base, extra = divmod(int(num_games), slots)
    if base + (1 if extra else 0) > 256:
        raise ValueError("more than 256 games per slot: raise max_slots (episode e of a slot runs on RNG stream + e * 2^24)")
    sp = make_self_play(config, network, slots, seed=seed, device=device, streams=streams, auto_reset=num_games > slots,
                        episodes=(base, extra) if num_games > slots else (0, 0))
    col = GameCollector(config, as_text=as_text, packed_only=packed_only) if rank == 0 else None
    nccl = multi and dist.get_backend() == "nccl"
    cdev = sp.device if nccl else None
    rec_bytes = _eng.RECORD_DTYPE.itemsize
    chunk = _chunk_ticks(config, slots)
    tower_split = getattr(network, "tower_mode", "") == "split"
    dev_index = sp.device.index or 0

    def as_records(t):
        return np.frombuffer(t.cpu().numpy().tobytes(), dtype=_eng.RECORD_DTYPE).copy()

    # Two chunks are in flight: the next chunk of ticks (and the drain behind it) is queued BEFORE the host waits for the
    # previous drain, so the device never idles for the host's turnaround; records and the counters that say "done" come
    # out of SelfPlay.take() on a side stream.  The price: one chunk of ticks is queued after the last game has ended
    # (every slot has stopped by then: its quota is played / auto_reset is off -- the ticks find nothing to do).
    sp.tick(chunk)
    pending, local_done, st = sp.drain_async(), False, None
    while True:
        nxt = None
        if not local_done:
            sp.tick(chunk)
            nxt = sp.drain_async()
        mine, st = sp.take(pending)                            # stays in HBM
        st["tower_overflow"] = _eng.tower_overflow(dev_index, reset=True) if tower_split else 0
        bad = int(bool(st["records_dropped"] or st["overflow"] or st["tower_overflow"]))
        local_done = st["games"] >= num_games
        if multi:
            if not nccl:
                mine = mine.cpu()
            info = parallel.all_gather_ints([mine.numel(), int(local_done), bad], dist, cdev)
            bad, all_done = int(info[:, 2].any()), bool(info[:, 1].all())
        else:
            all_done = local_done
        if bad:                                                # every rank leaves the loop together
            sp.synchronize()                                   # (the chunk queued ahead runs out first: nothing of this job
            if tower_split:                                    # is left on the device, nor in the range guard's counter)
                _eng.tower_overflow(dev_index, reset=True)
            raise _eng.EngineError("device record buffer or tree capacity overflowed, or (tower_overflow) an activation left "
                                   "the f16 range of the split tower kernels -- set network.tower_mode = 'fp32': %r" % (st,))
        if multi:
            parts = parallel.gather_tensor(mine, info[:, 0], dist) if info[:, 0].any() else None
            if rank == 0 and parts is not None:
                for rk, part in enumerate(parts):
                    if part.numel():
                        rr = as_records(part)
                        rr["slot"] += np.uint32(rk * slots)
                        col.add(rr[np.lexsort((rr["kind"], rr["ply"], rr["slot"], rr["episode"]))])
        else:
            r = as_records(mine)
            col.add(r[np.lexsort((r["kind"], r["ply"], r["slot"], r["episode"]))])
        if nxt is None:
            # this rank was done a round ago and has nothing queued; while other ranks still play it takes part in the
            # exchanges with an empty contribution
            pending = sp.drain_async()
        else:
            pending = nxt
        if all_done:
            break
    late, st = sp.take(pending)                                # queued after the last game ended: must be empty
    if late.numel():
        raise _eng.EngineError("%d bytes of records after the last game of the set" % late.numel())
    if rank != 0:
        return []
    games = col.finished
    games.sort(key=lambda g: (g["key"][1], g["key"][0]))
    if len(games) != num_games * world:
        raise _eng.EngineError("%d games finished, %d expected" % (len(games), num_games * world))
    return games


def games_from_records(config, recs, as_text=False):
    """raw records (possibly gathered from several ranks) -> finished games in (episode, slot) order;
    as_text: g["data"] holds each record as JSON text (bytes) instead of the 13-field list"""
    col = GameCollector(config, as_text=as_text)
    col.add(recs)
    col.finished.sort(key=lambda g: (g["key"][1], g["key"][0]))
    return col.finished


def _run_games(config, network, num_games, seed, device=0, max_slots=1024, streams=None, dist=None, as_text=False,
               packed_only=False):
    """Single process: all num_games here.  With an initialised torch.distributed group: rank r plays
    games [r * G/W, (r+1) * G/W) on its own GPU (disjoint RNG streams, no collective inside a search tick) and the
    records are gathered to rank 0 chunk by chunk while the games go on; rank 0 returns all games, other ranks []."""
    from . import parallel
    if parallel.active(dist):
        world, rank = dist.get_world_size(), dist.get_rank()
        if num_games % world:
            raise ValueError("num_games must be a multiple of the world size")
        per = num_games // world
        return _play_collect(config, network, per, seed, device=device, max_slots=max_slots,
                             streams=parallel.game_streams(rank, min(per, max_slots)), as_text=as_text, dist=dist), None
    return _play_collect(config, network, num_games, seed, device=device, max_slots=max_slots, streams=streams,
                         as_text=as_text, packed_only=packed_only), None


def _resolve_seed(seed, dist=None, device=0):
    """seed=None -> fresh randomness (fresh_seed); in a distributed job every rank gets rank 0's draw"""
    from . import parallel
    if seed is not None:
        return int(seed)
    seed = fresh_seed()
    if parallel.active(dist):
        import torch
        dev = torch.device("cuda", device) if dist.get_backend() == "nccl" else None
        seed = int(parallel.all_gather_ints([seed], dist, dev)[0, 0])
    return seed


def _log_seed(what, seed, **extra):
    """seeds that were drawn rather than given are written down, so that a set or a match can be replayed"""
    try:
        logs = storage_root() / "logs"
        logs.mkdir(parents=True, exist_ok=True)
        with open(logs / "seeds.jsonl", "a") as f:
            f.write(json.dumps(dict(what=what, seed=int(seed), **extra)) + "\n")
    except OSError:
        pass


def play_game(config, interference_network, game_number=None, screen=None, seed=None):
    """One self-play game -> (data, stats) as ai.py:1570-1699 (screen is ignored: no UI).  seed=None: fresh randomness
    as in the reference; with a seed the game is a function of (seed, game_number)."""
    games, _ = _run_games(config, interference_network, 1, _resolve_seed(seed),
                          streams=None if game_number is None else np.array([game_number], dtype=np.uint32))
    g = games[0]
    stats = {"model_number": highest_model_number(config), "model_version": config.model_version,
             "data_number": highest_data_number(config) + 1, "data_version": config.data_version,
             "app": g["stats"]["app"], "dspp": g["stats"]["dspp"]}
    return g["data"], stats


def make_training_set(config, interference_network, num_games, save_game=False, save_stats=True, screen=None, seed=None,
                      max_slots=1024, device=0, dist=None, write_json=True):
    """ai.py:1809-1845: plays num_games games, optionally writes {data_dir}/{n}.txt and the averaged
    stats line; returns the records only when save_stats is False (as the reference does).
    seed=None: fresh randomness per call, as the reference's (two calls never produce the same set); the drawn seed is
    logged in logs/seeds.jsonl.  An int makes the set reproducible: game g runs on RNG stream g of that seed.
    `dist`: an initialised torch.distributed module -> games are sharded over the ranks (one GPU each),
    records gathered to rank 0 over RCCL; files are written by rank 0 only.
    write_json=False (with save_game and save_stats): only the packed set is written -- the reference's JSON layout is
    ~60 KB of text per record and forming it is what bounds the host; this build's learner reads the packed set."""
    from . import parallel
    drawn = seed is None
    seed = _resolve_seed(seed, dist, device)
    # the records are only written, not returned (the reference's normal call): build them as JSON text directly --
    # the same bytes as json.dumps of the lists at a tenth of the host time (records.record_texts_for_move)
    as_text = bool(save_game and save_stats)
    packed_only = as_text and not write_json and not parallel.active(dist)
    games, _ = _run_games(config, interference_network, num_games, seed, max_slots=max_slots, device=device, dist=dist,
                          as_text=as_text, packed_only=packed_only)
    if parallel.active(dist) and dist.get_rank() != 0:
        return None
    series_data, series_stats = [], []
    for g in games:
        series_data.extend(g["data"])
        series_stats.append(g["stats"])
    if drawn and (save_game or save_stats):
        _log_seed("make_training_set", seed, data_number=highest_data_number(config) + 1, games=int(num_games))
    if save_game:
        os.makedirs(config.data_dir, exist_ok=True)
        next_set = highest_data_number(config) + 1
        if packed_only or (as_text and not write_json):
            pass
        elif as_text:
            with open("%s/%d.txt" % (config.data_dir, next_set), "wb") as f:
                f.write(b"[")
                for i, rec in enumerate(series_data):
                    f.write(rec if i == 0 else b", " + rec)
                f.write(b"]")
        else:
            with open("%s/%d.txt" % (config.data_dir, next_set), "w") as f:
                f.write(json.dumps(series_data))
        # the same saved moves in packed form for this build's learner (records.PackedSet; learner.packed_dir)
        from . import learner
        os.makedirs(learner.packed_dir(config), exist_ok=True)
        records.PackedSet.from_moves([p for g in games for p in g["packed"]], config.augment_data).save(
            "%s/%d.npz" % (learner.packed_dir(config), next_set))
    if save_stats:
        logs = storage_root() / "logs"
        logs.mkdir(parents=True, exist_ok=True)
        line = {"model_number": highest_model_number(config), "model_version": config.model_version,
                "data_number": highest_data_number(config) + (0 if save_game else 1), "data_version": config.data_version,
                "app": round(sum(s["app"] for s in series_stats) / len(series_stats), 3),
                "dspp": round(sum(s["dspp"] for s in series_stats) / len(series_stats), 3)}
        with open(logs / "stats.jsonl", "a") as f:
            f.write(json.dumps(line) + "\n")
        return None
    return series_data


def self_play_loop(config, skip_first_set=False, max_sets=None, train_fn=None, gate_fn=None, seed=None, write_json=True,
                   dist=None, device=0):
    """ai.py:2117-2198: generate a training set with the best model, train a challenger on the recent
    sets, play the gating match, promote the challenger if it meets the threshold -- forever, or
    `max_sets` times.  `train_fn(config, model)` / `gate_fn(challenger, config, best, config, threshold,
    threshold_type, games) -> (wins, accepted)` replace the built-in learner / gating match.

    seed=None: every set and every match on fresh randomness (the reference's behaviour).  With a base seed the seed of a
    set / a match is derived from (base, set number on disk, model number on disk), so a resumed loop continues with
    new games instead of replaying the first sets of the earlier run.

    `dist` (an initialised torch.distributed module, one rank per GPU): the games of a set and of the gating match are
    sharded over the ranks; rank 0 owns Storage/ -- it writes the sets, trains the challenger and broadcasts its weights
    (one flat bucket) before the match; the match's win vector is all-reduced, so every rank takes the same decision and
    ends each round with the same best network."""
    import inspect
    import torch
    from datetime import datetime
    from . import gating, learner, parallel
    multi = parallel.active(dist)
    rank = dist.get_rank() if multi else 0
    dev = torch.device("cuda", device) if torch.cuda.is_available() else torch.device("cpu")

    def barrier():
        if multi:
            dist.barrier()

    if rank == 0 and highest_model_number(config) == -1:
        instantiate_network(config, save_network=True)
    barrier()
    if rank == 0:
        best = load_best_model(config).to(dev)
    else:
        best = nets.model_for_config(config.model_config, config.use_tanh).to(dev)
    parallel.broadcast_state_dict(best, dist)            # every rank starts from rank 0's weights (Storage/ may be rank 0's only)
    training_config = config.copy()
    training_config.training = True
    logs = storage_root() / "logs"
    if rank == 0:
        logs.mkdir(parents=True, exist_ok=True)
    it = 0
    while max_sets is None or it < max_sets:
        it += 1
        challenger = nets.model_for_config(config.model_config, config.use_tanh).to(dev)
        challenger.load_state_dict(best.state_dict())
        if not skip_first_set:
            for i in range(config.training_loops):
                tag = parallel.all_gather_ints([highest_data_number(config) + 1, highest_model_number(config)], dist,
                                               dev if multi and dist.get_backend() == "nccl" else None)[0] if multi else \
                    (highest_data_number(config) + 1, highest_model_number(config))
                set_seed = None if seed is None else derive_seed(seed, "set", int(tag[0]), int(tag[1]))
                make_training_set(training_config, challenger, config.training_games, save_game=True, save_stats=True,
                                  seed=set_seed, write_json=write_json, dist=dist, device=device)
        else:
            skip_first_set = False
        next_ver = 0
        if rank == 0:
            (train_fn or learner.load_data_and_train_model)(config, challenger)
            next_ver = highest_model_number(config) + 1
        parallel.wait_for_rank0(dist)                      # training may outlast the backend's collective timeout: wait on the host
        parallel.broadcast_state_dict(challenger, dist)    # the trained challenger to every rank for the sharded match
        if hasattr(challenger, "_folded"):
            challenger._folded = None
        gate = gate_fn or gating.battle_networks
        kw = {}
        params = inspect.signature(gate).parameters
        if "seed" in params:
            tagm = (highest_data_number(config), highest_model_number(config))
            if multi:                                      # Storage/ may be rank 0's only: every rank derives the gate seed
                tagm = parallel.all_gather_ints(tagm, dist, dev if dist.get_backend() == "nccl" else None)[0]   # from rank 0's numbers
            kw["seed"] = None if seed is None else derive_seed(seed, "gate", int(tagm[0]), int(tagm[1]), it)
        if "dist" in params and multi:
            kw["dist"] = dist
        if "device" in params:
            kw["device"] = device
        win_loss, win = gate(challenger, config, best, config, config.gating_threshold, config.gating_threshold_type,
                             config.battle_games, **kw)
        if rank == 0:
            entry = {"model_version": config.model_version, "challenger_number": next_ver, "challenger_wins": int(win_loss[0]),
                     "best_wins": int(win_loss[1]), "total_games": config.battle_games,
                     "win_rate": float(win_loss[0]) / config.battle_games, "accepted": bool(win)}
            with open(logs / "gating_log.jsonl", "a") as f:
                f.write(json.dumps(entry) + "\n")
        if win:
            if rank == 0:
                torch.save(challenger.state_dict(), "%s/%d.pt" % (config.model_dir, next_ver))
                with open(Path(config.model_dir) / "versions.jsonl", "a") as f:       # append_version_record, ai.py:224-234
                    f.write(json.dumps({"timestamp": datetime.utcnow().isoformat() + "Z", "model_number": next_ver,
                                        "backend": config.model}) + "\n")
            best = challenger
        barrier()
    global _last_best
    _last_best = best                                      # the network this process holds after the loop (tests)
    return it


_last_best = None
