"""Gating match on the same engine: battle_networks of the reference
(ai.py:1975-2114) in both of its forms -- the concurrent one (_battle_networks_async: all games at once,
network 1 plays player `i % 2` of game i, no early termination, threshold checked afterwards) and the
sequential one (ai.py:1996-2049: games in order, stop as soon as either network meets the threshold).
On the device the games of either form run concurrently; the sequential form's early return is
reproduced by scanning the finished games in index order (game i runs on RNG stream i either way, so
its result does not depend on when it is played) and by not starting later waves once the scan stops.

Every slot's search is evaluated by the network that owns the side to move at the ROOT of that
search; both networks evaluate the whole leaf batch and the results are blended per slot on the
device (no host round trip inside a tick).
"""
import ctypes as C

import numpy as np

from . import engine as _eng
from . import selfplay


class _DevView:
    """raw device memory as a torch tensor (zero copy) through __cuda_array_interface__"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}


def _check_threshold(wins, games, threshold, threshold_type):
    """The gate of ai.py:2031-2046 / 2055-2069 as one rule: with 'more' the challenger needs strictly more than
    threshold * games wins and is rejected once the holder has (1 - threshold) * games; with 'moreorequal' the two
    comparisons swap strictness.  True = accepted, False = rejected, None = undecided (or no / unknown rule)."""
    strict = {"more": True, "moreorequal": False}.get(threshold_type)
    if threshold is None or strict is None:
        return None
    need, against = threshold * games, (1 - threshold) * games
    if (wins[0] > need) if strict else (wins[0] >= need):
        return True
    if (wins[1] >= against) if strict else (wins[1] > against):
        return False
    return None


class Battle:
    """`games` concurrent games between two evaluators; evaluator k is a callable
    (grids, vec) -> (values [n], policy [n, 11583]) on the device -- or, with `nets=(net_1, net_2)` two AlphaSame
    nets that run on the fused tower: then a tick is select -> tower of net 1 on its slots' flagged boards -> tower
    of net 2 on the others -> one expand launch per net (dense heads inside the kernel), one board per leaf as in
    self-play, instead of both nets' full forward pass over every slot."""

    def __init__(self, config, eval_1, eval_2, games, seed=0, device=0, streams=None, nets=None, index_base=0):
        """index_base: global index of this Battle's first game (waves of the sequential form, ranks of a sharded
        match): game k here is game index_base + k of the match -- its colour and its default RNG stream follow that"""
        import torch
        self.torch = torch
        self.n = int(games)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.engine = _eng.Engine(self.n, ruleset=config.ruleset, seed=seed, device=device, stream=self.stream.cuda_stream)
        self.index_base = int(index_base)
        if streams is None and index_base:
            streams = np.arange(index_base, index_base + self.n, dtype=np.uint32)
        self.engine.reset(streams)
        # only play_game opens with random moves (ai.py:1588-1608): a gating match searches from ply 0
        self.engine.mcts_create(selfplay.search_options(config, auto_reset=0, random_start=0))
        self.evals = (eval_1, eval_2)
        for ev in self.evals:
            if hasattr(ev, "bind"):
                ev.bind(self)                             # evaluators that need the engine (the tests' stub evaluators)
        self.nets = None
        if nets is not None and all(getattr(m, "supports_tower_cache", False) and hasattr(m, "device_heads") for m in nets):
            heads = [m.device_heads() for m in nets]
            if all(h is not None for h in heads):
                self.nets, self.heads = tuple(nets), heads
        with torch.cuda.stream(self.stream):
            self.grids = torch.zeros((2, self.n, _eng.GRID_FLOATS), dtype=torch.float32, device=self.device)
            self.vec = torch.zeros((self.n, _eng.VEC_FLOATS), dtype=torch.float32, device=self.device)
            self.side = ((torch.arange(self.n, device=self.device) + self.index_base) % 2).to(torch.uint8)   # network 1 plays player i % 2
            self.states = torch.as_tensor(_DevView(self.engine.state_dev_ptr(), (self.n, _eng.STATE_BYTES), "|u1"),
                                          device=self.device)
            if self.nets is not None:
                self.need = torch.ones(2 * self.n, dtype=torch.uint8, device=self.device)
                self.z = torch.zeros((2 * self.n, _eng.GRID_FLOATS), dtype=torch.float32, device=self.device)
        self.stream.synchronize()

    def _tick_native(self):
        torch, e = self.torch, self.engine
        e.mcts_select_need(self.grids.data_ptr(), self.vec.data_ptr(), self.need.data_ptr())
        first = (self.states[:, 312] == self.side).to(torch.uint8)       # root side to move is network 1's colour
        need1 = self.need * first.repeat(2)
        need2 = self.need - need1
        self.nets[0].tower_need(self.grids, need1, out=self.z)
        self.nets[1].tower_need(self.grids, need2, out=self.z)
        e.mcts_expand_alphasame_masked(self.z.data_ptr(), self.vec.data_ptr(), self.heads[0], first.data_ptr(), 1, 0)
        e.mcts_expand_alphasame_masked(self.z.data_ptr(), self.vec.data_ptr(), self.heads[1], first.data_ptr(), 0, 1)
        self._keep = (first, need1, need2)                                # alive until the stream has used them

    def tick(self, n=1):
        torch = self.torch
        e = self.engine
        with torch.cuda.stream(self.stream):
            if self.nets is not None:
                for _ in range(n):
                    self._tick_native()
                return
            for _ in range(n):
                e.mcts_select(self.grids.data_ptr(), self.vec.data_ptr())
                first = self.states[:, 312] == self.side               # root side to move is network 1's colour
                v1, p1 = self.evals[0](self.grids, self.vec)
                v2, p2 = self.evals[1](self.grids, self.vec)
                v = torch.where(first, v1, v2).contiguous()
                p = torch.where(first[:, None], p1, p2).contiguous()
                e.mcts_expand(v.data_ptr(), p.data_ptr())

    def run(self, max_ticks=10 ** 9, chunk=64):
        """plays all games to the end; returns wins [network 1, network 2] (draw = half a point each);
        self.winners[k] = Game.winner of game k, self.moves[k] = [(ply, move)]"""
        wins = np.zeros(2, dtype=float)
        ended, t = 0, 0
        self.moves, self.winners = {}, {}
        while ended < self.n and t < max_ticks:
            self.tick(chunk)
            t += chunk
            self.stream.synchronize()
            recs = self.engine.mcts_drain_records()
            for r in recs:
                if r["kind"] == _eng.REC_MOVE:
                    self.moves.setdefault(int(r["slot"]), []).append((int(r["ply"]), int(r["chosen"])))
                else:
                    ended += 1
                    self.winners[int(r["slot"])] = int(r["winner"])
        st = self.engine.mcts_stats()
        if self.nets is not None and any(getattr(m, "tower_mode", "") == "split" for m in self.nets):
            st["tower_overflow"] = _eng.tower_overflow(self.device.index or 0, reset=True)     # the stream is synchronised
        if st["records_dropped"] or st["overflow"] or st.get("tower_overflow"):
            raise _eng.EngineError("device record buffer or tree capacity overflowed, or (tower_overflow) an activation left "
                                   "the f16 range of the split tower kernels -- set tower_mode = 'fp32' on the networks: %r" % (st,))
        return tally_concurrent([self.winners[k] for k in range(self.n)], self.index_base) if ended == self.n else wins


def tally_concurrent(winners, first=0):
    """_battle_networks_async's tally (ai.py:2105-2113): winners[k] = Game.winner of game first + k"""
    wins = np.zeros(2, dtype=float)
    for k, winner in enumerate(winners):
        side = (first + k) % 2
        if winner == -1:
            wins += 0.5
        elif side == 0:
            wins[winner] += 1
        else:
            wins[1 - winner] += 1
    return wins


def scan_sequential(wins, winners, first, games, threshold, threshold_type):
    """The sequential loop's bookkeeping (ai.py:2019-2046) over games first, first + 1, ... whose winners are known:
    updates the float `wins` in place and returns (games consumed, True / False when a threshold was met, else None).
    A drawn game (MAX_MOVES reached) counts half a point for each network; the reference's own `wins += 0.5` on its
    integer array raises numpy's UFuncTypeError there (ai.py:2021), which would abort self_play_loop after the set was
    generated and the challenger trained -- not reproduced."""
    for k, winner in enumerate(winners):
        flip_color = (first + k) % 2 == 1
        if winner == -1:
            wins += 0.5
        elif not flip_color:
            wins[winner] += 1
        else:
            wins[1 - winner] += 1
        accepted = _check_threshold(wins, games, threshold, threshold_type)
        if accepted is not None:
            return k + 1, accepted
    return len(winners), None


def _as_reference_wins(wins):
    """the sequential form's result array: int64 like the reference's when every game had a winner"""
    return wins.astype(int) if np.all(wins == np.floor(wins)) else wins


def _fresh_seed():
    """the reference plays every match on fresh randomness (CPython's global generator): draw a seed from the OS"""
    import os
    return int.from_bytes(os.urandom(8), "little") >> 1


def battle_networks(NN_1, config_1, NN_2, config_2, threshold, threshold_type, games, network_1_title="Network 1",
                    network_2_title="Network 2", screen=None, seed=None, device=0, dist=None, wave=64, evaluators=None):
    """ai.py:1975-2052 -> (wins array, True / False / None).
    Which form runs follows the reference: both configs model == 'pytorch' with batched_inference -> the concurrent form
    (_battle_networks_async + _check_threshold, float wins); otherwise the sequential form: integer wins, games in index
    order, return at the first game after which either network meets the threshold (ai.py:2031-2046) -- played in waves of
    `wave` concurrent games, later waves are not started once the scan has stopped.  (A drawn game counts half a point
    each: the wins come back as integers when no game was drawn -- the reference's array -- and as floats otherwise,
    where the reference's `wins += 0.5` on its integer array raises.)
    seed=None: fresh randomness per match, as the reference's; an int makes the match reproducible (game i = stream i).
    dist: an initialised torch.distributed module -> rank r plays games [r * G / W, (r + 1) * G / W), the winners are
    all-gathered and every rank returns the same result (either form; no wave savings when sharded).
    evaluators: (eval_1, eval_2) callables replacing the networks' evaluate_packed (tests: stub evaluators)."""
    if config_1.ruleset != config_2.ruleset:
        raise NotImplementedError("Ruleset's aren't equal")
    o1, o2 = selfplay.search_options(config_1), selfplay.search_options(config_2)
    if bytes(o1) != bytes(o2):
        raise NotImplementedError("the device runs one set of search options for all slots")
    import torch
    from . import parallel
    dev = torch.device("cuda", device)
    multi = parallel.active(dist)
    if seed is None:
        seed = _fresh_seed()
        if multi:                                     # every rank plays the same match
            seed = int(parallel.all_gather_ints([seed], dist, dev if dist.get_backend() == "nccl" else None)[0, 0])
    if evaluators is None:
        n1, n2 = NN_1.to(dev).prepare(), NN_2.to(dev).prepare()
        evaluators, nets = (n1.evaluate_packed, n2.evaluate_packed), (n1, n2)
    else:
        nets = None

    def play(first, count):
        b = Battle(config_1, evaluators[0], evaluators[1], count, seed=seed, device=device, nets=nets, index_base=first)
        b.run()
        return [b.winners[k] for k in range(count)]

    use_async = (config_1.model == "pytorch" and config_2.model == "pytorch"
                 and config_1.batched_inference and config_2.batched_inference)
    if multi:
        # games are independent: rank r plays games [r * G / W, (r + 1) * G / W); the winners are all-gathered, then every
        # rank applies the form's bookkeeping to the whole list -- the sequential form's early return included
        world, rank = dist.get_world_size(), dist.get_rank()
        if games % world:
            raise ValueError("games must be a multiple of the world size")
        per = games // world
        mine = play(rank * per, per)
        allw = parallel.all_gather_ints([w + 1 for w in mine], dist, dev if dist.get_backend() == "nccl" else None)
        winners = [int(w) - 1 for w in allw.reshape(-1)]
        if use_async:
            wins = tally_concurrent(winners)
            return wins, _check_threshold(wins, games, threshold, threshold_type)
        wins = np.zeros(2, dtype=float)
        _, accepted = scan_sequential(wins, winners, 0, games, threshold, threshold_type)
        return _as_reference_wins(wins), accepted
    if use_async:
        wins = tally_concurrent(play(0, games))
        return wins, _check_threshold(wins, games, threshold, threshold_type)

    wins = np.zeros(2, dtype=float)                       # ai.py:1991 (integers there)
    i = 0
    while i < games:
        count = min(int(wave), games - i)
        _, accepted = scan_sequential(wins, play(i, count), i, games, threshold, threshold_type)
        if accepted is not None:
            return _as_reference_wins(wins), accepted
        i += count
    return _as_reference_wins(wins), None
