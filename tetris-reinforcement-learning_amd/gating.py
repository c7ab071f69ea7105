"""Gating match on the same engine: battle_networks of the reference
(ai.py:1975-2114) in its concurrent form (_battle_networks_async: all games at once, network 1 plays
player `i % 2` of game i, no early termination, threshold checked afterwards).

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
    """ai.py:2055-2069"""
    if threshold is None:
        return None
    if threshold_type == "more":
        if wins[0] > threshold * games:
            return True
        if wins[1] >= (1 - threshold) * games:
            return False
    elif threshold_type == "moreorequal":
        if wins[0] >= threshold * games:
            return True
        if wins[1] > (1 - threshold) * games:
            return False
    return None


class Battle:
    """`games` concurrent games between two evaluators; evaluator k is a callable
    (grids, vec) -> (values [n], policy [n, 11583]) on the device -- or, with `nets=(net_1, net_2)` two AlphaSame
    nets that run on the fused tower: then a tick is select -> tower of net 1 on its slots' flagged boards -> tower
    of net 2 on the others -> one expand launch per net (dense heads inside the kernel), one board per leaf as in
    self-play, instead of both nets' full forward pass over every slot."""

    def __init__(self, config, eval_1, eval_2, games, seed=0, device=0, streams=None, nets=None):
        import torch
        self.torch = torch
        self.n = int(games)
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.engine = _eng.Engine(self.n, ruleset=config.ruleset, seed=seed, device=device, stream=self.stream.cuda_stream)
        self.engine.reset(streams)
        self.engine.mcts_create(selfplay.search_options(config, auto_reset=0))
        self.evals = (eval_1, eval_2)
        self.nets = None
        if nets is not None and all(getattr(m, "supports_tower_cache", False) and hasattr(m, "device_heads") for m in nets):
            heads = [m.device_heads() for m in nets]
            if all(h is not None for h in heads):
                self.nets, self.heads = tuple(nets), heads
        with torch.cuda.stream(self.stream):
            self.grids = torch.zeros((2, self.n, _eng.GRID_FLOATS), dtype=torch.float32, device=self.device)
            self.vec = torch.zeros((self.n, _eng.VEC_FLOATS), dtype=torch.float32, device=self.device)
            self.side = (torch.arange(self.n, device=self.device) % 2).to(torch.uint8)       # network 1 plays player i % 2
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
        """plays all games to the end; returns wins [network 1, network 2] (draw = half a point each)"""
        wins = np.zeros(2, dtype=float)
        ended, t = 0, 0
        self.moves = {}
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
                    side, winner = int(r["slot"]) % 2, int(r["winner"])
                    if winner == -1:
                        wins += 0.5
                    elif side == 0:
                        wins[winner] += 1
                    else:
                        wins[1 - winner] += 1
        st = self.engine.mcts_stats()
        if st["records_dropped"] or st["overflow"]:
            raise _eng.EngineError("device record buffer or tree capacity overflowed: %r" % (st,))
        return wins


def battle_networks(NN_1, config_1, NN_2, config_2, threshold, threshold_type, games, network_1_title="Network 1",
                    network_2_title="Network 2", screen=None, seed=0, device=0):
    """ai.py:1975-2052 -> (wins array, True / False / None).  Concurrent form only (see module doc)."""
    if config_1.ruleset != config_2.ruleset:
        raise NotImplementedError("Ruleset's aren't equal")
    o1, o2 = selfplay.search_options(config_1), selfplay.search_options(config_2)
    if bytes(o1) != bytes(o2):
        raise NotImplementedError("the device runs one set of search options for all slots")
    import torch
    dev = torch.device("cuda", device)
    n1, n2 = NN_1.to(dev).prepare(), NN_2.to(dev).prepare()
    b = Battle(config_1, n1.evaluate_packed, n2.evaluate_packed, games, seed=seed, device=device, nets=(n1, n2))
    wins = b.run()
    return wins, _check_threshold(wins, games, threshold, threshold_type)
