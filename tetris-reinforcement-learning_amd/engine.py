"""ctypes binding of include/tetris_engine.h (libtrl_hip.so).

There is no CPU fallback: if the shared library is missing or no gfx950 device is visible,
constructing an Engine raises.  Nothing here imports oracle/.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

STATE_BYTES = 320
MAX_LEGAL = 512
MOVE_NONE = 0xFFFF
RULESETS = {"s1": 1, "s2": 2}
MOVE_ALGORITHMS = {"convolutional": 0, "brute-force": 1}

# every symbol include/tetris_engine.h declares (tests/test_abi.py checks the header against this)
ABI_SYMBOLS = [
    "trl_engine_create", "trl_engine_destroy", "trl_set_stream", "trl_last_error", "trl_version",
    "trl_env_reset", "trl_env_get_state", "trl_env_set_state", "trl_env_state_dev",
    "trl_env_legal_moves", "trl_env_legal_moves_algo", "trl_env_legal_dev", "trl_env_step", "trl_env_step_dev",
    "trl_env_rollout_random", "trl_last_kernel_ms",
    "trl_mcts_create", "trl_mcts_select", "trl_mcts_expand", "trl_mcts_expand_head", "trl_mcts_drain_records", "trl_mcts_drain_records_dev", "trl_mcts_drain_records_async", "trl_mcts_get_stats",
    "trl_mcts_leaf_states", "trl_mcts_root_children", "trl_alphasame_trunk16", "trl_alphasame_trunk16_need", "trl_alphasame_trunk16_split", "trl_trunk_wide_split", "trl_alphasame_trunk", "trl_resnet_trunk", "trl_graph_begin", "trl_graph_end", "trl_graph_end_many", "trl_graph_launch",
    "trl_tower_overflow",
    "trl_mcts_select_need", "trl_mcts_fork_legal", "trl_mcts_expand_alphasame_masked", "trl_mcts_zcache", "trl_mcts_zcache_dim", "trl_mcts_expand_alphasame", "trl_mcts_expand_alphasame_select", "trl_mcts_restart", "trl_mcts_legal_in_tower", "trl_mcts_tower16_legal",
]

GRID_FLOATS = 400
VEC_FLOATS = 105
POLICY_SIZE = 27 * 39 * 11


class TrlMctsConfig(C.Structure):
    """trl_mcts_config (include/tetris_engine.h); defaults = Config() of the reference, ai.py:63-138"""
    _fields_ = [("max_iter", C.c_int32), ("training", C.c_int32), ("fpu_reduction", C.c_int32),
                ("use_root_softmax", C.c_int32), ("use_playout_cap", C.c_int32), ("use_dirichlet", C.c_int32),
                ("use_dir_s", C.c_int32), ("use_forced", C.c_int32), ("save_all", C.c_int32),
                ("max_nodes", C.c_int32), ("record_capacity", C.c_int32), ("auto_reset", C.c_int32),
                ("use_tanh", C.c_int32), ("random_start", C.c_int32), ("episodes_base", C.c_int32),
                ("episodes_extra", C.c_int32), ("move_algorithm", C.c_int32), ("reserved0", C.c_int32), ("cpuct", C.c_double), ("dpuct", C.c_double), ("fpu_value", C.c_double),
                ("root_softmax_temp", C.c_double), ("temperature", C.c_double),
                ("playout_cap_chance", C.c_double), ("playout_cap_mult", C.c_double),
                ("dir_alpha", C.c_double), ("dir_s", C.c_double), ("dir_expl", C.c_double), ("c_forced", C.c_double)]


def mcts_config(**kw):
    c = TrlMctsConfig(max_iter=400, training=0, fpu_reduction=1, use_root_softmax=1, use_playout_cap=1,
                      use_dirichlet=1, use_dir_s=1, use_forced=0, save_all=0, max_nodes=0, record_capacity=0,
                      auto_reset=1, use_tanh=0, random_start=0, episodes_base=0, episodes_extra=0, move_algorithm=0, reserved0=0, cpuct=0.75, dpuct=1.0, fpu_value=0.1, root_softmax_temp=1.1, temperature=0.1,
                      playout_cap_chance=0.25, playout_cap_mult=5.0, dir_alpha=0.1, dir_s=25.0, dir_expl=0.25,
                      c_forced=1.0)
    for k, v in kw.items():
        if not hasattr(c, k):
            raise TypeError("unknown search option %r" % k)
        setattr(c, k, v)
    return c


RECORD_DTYPE = np.dtype([("slot", "<u4"), ("episode", "<u4"), ("ply", "<u4"), ("n_children", "<u2"), ("kind", "u1"),
                         ("winner", "i1"), ("chosen", "<u2"), ("saved", "u1"), ("random_move", "u1"), ("pad2_", "<u4"),
                         ("state", "u1", (320,)), ("moves", "<u2", (512,)), ("visits", "<u2", (512,))])
assert RECORD_DTYPE.itemsize == 2392
REC_MOVE, REC_GAME_END = 0, 1


class TrlAlphaSameHeads(C.Structure):
    _fields_ = [("oside_w", C.c_void_p), ("oside_b", C.c_void_p), ("value0_w", C.c_void_p), ("value0_b", C.c_void_p),
                ("value3_w", C.c_void_p), ("policy_w", C.c_void_p), ("policy_b", C.c_void_p), ("value3_b", C.c_float),
                ("use_tanh", C.c_int32)]


class TrlMctsStats(C.Structure):
    _fields_ = [("ticks", C.c_uint64), ("simulations", C.c_uint64), ("moves", C.c_uint64), ("games", C.c_uint64),
                ("records_dropped", C.c_uint64), ("max_nodes_used", C.c_uint32), ("overflow", C.c_uint32),
                ("boards_reused", C.c_uint64)]



class TrlConfig(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("ruleset", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("seed", C.c_uint64), ("stream", C.c_void_p)]


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libtrl_hip.so and set prototypes.  Nothing is compiled here: the library is built by
    tetris_rl_amd.build() / __graft_entry__.build() (hipcc, in-tree) and a missing one is an error -- the GPU box runs
    the .so that was built with the sources it ships with."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _build.LIB_PATH
    # PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64; two HSA runtimes in one process cannot
    # both open the device.  Loading torch first makes this library bind to the runtime torch uses
    # (same soname), so engine buffers and torch tensors share one HIP context.
    import torch  # noqa: F401
    if not os.path.isfile(path):
        raise EngineError("HIP engine library %s is missing: run `python __graft_entry__.py` (build()) first; "
                          "there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.trl_last_error.restype = C.c_char_p
    L.trl_version.restype = C.c_char_p
    L.trl_engine_create.argtypes = [C.POINTER(TrlConfig), C.POINTER(C.c_void_p)]
    L.trl_engine_destroy.argtypes = [C.c_void_p]
    L.trl_engine_destroy.restype = None
    L.trl_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.trl_env_reset.argtypes = [C.c_void_p, C.c_void_p]
    L.trl_env_get_state.argtypes = [C.c_void_p, C.c_void_p]
    L.trl_env_set_state.argtypes = [C.c_void_p, C.c_void_p]
    L.trl_env_state_dev.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.trl_env_legal_moves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_env_legal_moves_algo.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.trl_env_legal_dev.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.trl_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.trl_env_step_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.trl_env_rollout_random.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.trl_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.trl_mcts_create.argtypes = [C.c_void_p, C.POINTER(TrlMctsConfig)]
    L.trl_mcts_select.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_mcts_expand.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_mcts_expand_head.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.trl_mcts_drain_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.trl_mcts_drain_records_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    L.trl_mcts_drain_records_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
    L.trl_mcts_get_stats.argtypes = [C.c_void_p, C.POINTER(TrlMctsStats)]
    L.trl_mcts_leaf_states.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_mcts_root_children.argtypes = [C.c_void_p] + [C.c_void_p] * 5
    L.trl_mcts_select_need.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_mcts_fork_legal.argtypes = [C.c_void_p, C.c_int]
    L.trl_mcts_zcache.argtypes = [C.c_void_p, C.c_void_p]
    L.trl_mcts_zcache_dim.argtypes = [C.c_void_p, C.c_int]
    L.trl_graph_begin.argtypes = [C.c_void_p]
    L.trl_graph_end.argtypes = [C.c_void_p]
    L.trl_graph_end_many.argtypes = [C.c_void_p, C.c_int]
    L.trl_graph_launch.argtypes = [C.c_void_p, C.c_int]
    L.trl_mcts_expand_alphasame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(TrlAlphaSameHeads)]
    L.trl_mcts_restart.argtypes = [C.c_void_p]
    L.trl_mcts_legal_in_tower.argtypes = [C.c_void_p, C.c_int]
    L.trl_mcts_tower16_legal.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.trl_mcts_expand_alphasame_select.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(TrlAlphaSameHeads), C.c_void_p, C.c_void_p]
    L.trl_mcts_expand_alphasame_masked.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(TrlAlphaSameHeads),
                                                   C.c_void_p, C.c_int, C.c_int]
    L.trl_alphasame_trunk16_need.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_alphasame_trunk16_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_trunk_wide_split.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    L.trl_alphasame_trunk16.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.trl_alphasame_trunk.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_resnet_trunk.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.trl_tower_overflow.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    _lib = L
    return L


def tower_overflow(device=0, reset=True):
    """boards whose activations left the f16 range inside a split tower since the last reset (trl_tower_overflow); the
    caller has synchronised the streams its towers ran on"""
    L = load_library()
    n = C.c_uint32(0)
    _check(L, L.trl_tower_overflow(int(device), int(bool(reset)), C.byref(n)), "trl_tower_overflow")
    return int(n.value)


def _check(L, rc, what):
    if rc != 0:
        raise EngineError("%s failed (%d): %s" % (what, rc, L.trl_last_error().decode()))


class Engine:
    """n_games concurrent two-player games resident in HBM.

    Mirrors the reference's env seam S8 in batch form:
    Game(ruleset).setup() -> reset(); get_move_matrix/get_move_list -> legal_moves();
    Game.make_move -> step().
    """

    def __init__(self, n_games, ruleset="s2", seed=0, device=0, stream=None):
        if ruleset not in RULESETS:
            raise ValueError("Unknown ruleset: %r" % (ruleset,))
        self._L = load_library()
        self.n_games = int(n_games)
        self.ruleset = ruleset
        self.seed = int(seed)
        cfg = TrlConfig(self.n_games, RULESETS[ruleset], device, 0, self.seed, stream)
        self._h = C.c_void_p()
        self.state_epoch = 0            # counts the calls that replace or advance the game states from outside the search
        _check(self._L, self._L.trl_engine_create(C.byref(cfg), C.byref(self._h)), "trl_engine_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.trl_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- env ---------------------------------------------------------------------------------
    def reset(self, streams=None):
        sp = None
        if streams is not None:
            streams = np.ascontiguousarray(streams, dtype=np.uint32)
            if streams.shape != (self.n_games,):
                raise ValueError("streams must have shape (n_games,)")
            sp = streams.ctypes.data
        _check(self._L, self._L.trl_env_reset(self._h, sp), "trl_env_reset")
        self.state_epoch += 1

    def get_state(self):
        out = np.empty((self.n_games, STATE_BYTES), dtype=np.uint8)
        _check(self._L, self._L.trl_env_get_state(self._h, out.ctypes.data), "trl_env_get_state")
        return out

    def set_state(self, states):
        states = np.ascontiguousarray(states, dtype=np.uint8)
        if states.shape != (self.n_games, STATE_BYTES):
            raise ValueError("states must have shape (n_games, %d)" % STATE_BYTES)
        _check(self._L, self._L.trl_env_set_state(self._h, states.ctypes.data), "trl_env_set_state")
        self.state_epoch += 1

    def legal_moves(self, algo="convolutional"):
        """algo: 'convolutional' (Config.move_algorithm's default, what the search uses) or 'brute-force'
        (get_move_matrix's default argument); anything else raises ValueError as the reference does"""
        if algo not in MOVE_ALGORITHMS:
            raise ValueError("Unknown algorithm: %s" % algo)
        counts = np.empty(self.n_games, dtype=np.uint16)
        moves = np.empty((self.n_games, MAX_LEGAL), dtype=np.uint16)
        _check(self._L, self._L.trl_env_legal_moves_algo(self._h, MOVE_ALGORITHMS[algo], counts.ctypes.data, moves.ctypes.data),
               "trl_env_legal_moves_algo")
        return counts, moves

    def step(self, moves, add_bag=True, check_legal=True):
        moves = np.ascontiguousarray(moves, dtype=np.uint16)
        if moves.shape != (self.n_games,):
            raise ValueError("moves must have shape (n_games,)")
        _check(self._L, self._L.trl_env_step(self._h, moves.ctypes.data, int(bool(add_bag)), int(bool(check_legal))),
               "trl_env_step")
        self.state_epoch += 1

    def rollout_random(self, n_plies, max_plies=2000, sync=True):
        steps = C.c_uint64(0)
        _check(self._L, self._L.trl_env_rollout_random(self._h, int(n_plies), int(max_plies),
                                                        C.byref(steps) if sync else None), "trl_env_rollout_random")
        self.state_epoch += 1
        return steps.value if sync else None

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(self._L, self._L.trl_last_kernel_ms(self._h, C.byref(ms)), "trl_last_kernel_ms")
        return ms.value

    def state_dev_ptr(self):
        p = C.c_void_p()
        _check(self._L, self._L.trl_env_state_dev(self._h, C.byref(p)), "trl_env_state_dev")
        return p.value

    def set_stream(self, hip_stream):
        _check(self._L, self._L.trl_set_stream(self._h, hip_stream), "trl_set_stream")

    # -- search ------------------------------------------------------------------------------
    def mcts_create(self, cfg):
        self._rec_cap = cfg.record_capacity if cfg.record_capacity > 0 else 16384
        _check(self._L, self._L.trl_mcts_create(self._h, C.byref(cfg)), "trl_mcts_create")
        # persistent page-locked staging buffer for record drains (allocating / faulting it per drain costs more
        # than the copy)
        import torch
        self._rec_pinned = torch.empty(self._rec_cap * RECORD_DTYPE.itemsize, dtype=torch.uint8, pin_memory=True)
        self._rec_buf = self._rec_pinned.numpy().view(RECORD_DTYPE)

    def mcts_select(self, grids_ptr, vec_ptr):
        _check(self._L, self._L.trl_mcts_select(self._h, grids_ptr, vec_ptr), "trl_mcts_select")

    def mcts_select_need(self, grids_ptr, vec_ptr, need_ptr):
        _check(self._L, self._L.trl_mcts_select_need(self._h, grids_ptr, vec_ptr, need_ptr), "trl_mcts_select_need")

    def mcts_fork_legal(self, on=True):
        _check(self._L, self._L.trl_mcts_fork_legal(self._h, int(bool(on))), "trl_mcts_fork_legal")

    def graph_begin(self):
        _check(self._L, self._L.trl_graph_begin(self._h), "trl_graph_begin")

    def graph_end(self):
        _check(self._L, self._L.trl_graph_end(self._h), "trl_graph_end")

    def graph_end_many(self, ticks):
        _check(self._L, self._L.trl_graph_end_many(self._h, int(ticks)), "trl_graph_end_many")

    def graph_launch(self, n=1):
        _check(self._L, self._L.trl_graph_launch(self._h, int(n)), "trl_graph_launch")

    def mcts_zcache_dim(self, floats_per_board):
        _check(self._L, self._L.trl_mcts_zcache_dim(self._h, int(floats_per_board)), "trl_mcts_zcache_dim")

    def mcts_zcache(self, z_ptr):
        _check(self._L, self._L.trl_mcts_zcache(self._h, z_ptr), "trl_mcts_zcache")

    def mcts_expand_alphasame(self, z_ptr, vec_ptr, heads):
        _check(self._L, self._L.trl_mcts_expand_alphasame(self._h, z_ptr, vec_ptr, C.byref(heads)), "trl_mcts_expand_alphasame")

    def mcts_legal_in_tower(self, on=True):
        """the leaf's legal-move search as extra workgroups of the tower launch (trl_mcts_tower16_legal) instead of the
        selection's"""
        _check(self._L, self._L.trl_mcts_legal_in_tower(self._h, 1 if on else 0), "trl_mcts_legal_in_tower")

    def mcts_tower16_legal(self, boards_ptr, n_boards, packed_ptr, n_blocks, out_ptr, need_ptr):
        _check(self._L, self._L.trl_mcts_tower16_legal(self._h, boards_ptr, int(n_boards), packed_ptr, int(n_blocks), out_ptr, need_ptr),
               "trl_mcts_tower16_legal")

    def mcts_restart(self):
        _check(self._L, self._L.trl_mcts_restart(self._h), "trl_mcts_restart")

    def mcts_expand_alphasame_select(self, z_ptr, vec_ptr, heads, grids_ptr, need_ptr):
        _check(self._L, self._L.trl_mcts_expand_alphasame_select(self._h, z_ptr, vec_ptr, C.byref(heads), grids_ptr, need_ptr),
               "trl_mcts_expand_alphasame_select")

    def mcts_expand_alphasame_masked(self, z_ptr, vec_ptr, heads, mask_ptr, want, last):
        _check(self._L, self._L.trl_mcts_expand_alphasame_masked(self._h, z_ptr, vec_ptr, C.byref(heads), mask_ptr, int(want), int(last)),
               "trl_mcts_expand_alphasame_masked")

    def mcts_expand(self, values_ptr, policy_ptr):
        _check(self._L, self._L.trl_mcts_expand(self._h, values_ptr, policy_ptr), "trl_mcts_expand")

    def mcts_expand_head(self, values_ptr, feats_ptr, feat_dim, weight_ptr, bias_ptr):
        _check(self._L, self._L.trl_mcts_expand_head(self._h, values_ptr, feats_ptr, int(feat_dim), weight_ptr, bias_ptr),
               "trl_mcts_expand_head")

    def mcts_drain_records(self):
        buf = self._rec_buf
        n = C.c_int(0)
        _check(self._L, self._L.trl_mcts_drain_records(self._h, buf.ctypes.data, self._rec_cap, C.byref(n)),
               "trl_mcts_drain_records")
        return buf[:n.value].copy()

    def mcts_drain_records_dev(self, dev_ptr, cap):
        """pending records -> device buffer at dev_ptr (room for `cap` records); returns the count"""
        n = C.c_int(0)
        _check(self._L, self._L.trl_mcts_drain_records_dev(self._h, dev_ptr, int(cap), C.byref(n)), "trl_mcts_drain_records_dev")
        return n.value

    def mcts_drain_records_async(self, dev_ptr, cap, head_ptr, slot_offset=0):
        """pending records appended to the device buffer at dev_ptr (room for `cap` records in all) behind the head_ptr[0]
        records already there; head_ptr: uint64[9] on the device ([0] count, [1..8] search counters at this drain).
        Queued on the engine's stream; nothing is waited for."""
        _check(self._L, self._L.trl_mcts_drain_records_async(self._h, dev_ptr, int(cap), head_ptr, int(slot_offset)),
               "trl_mcts_drain_records_async")

    def mcts_stats(self):
        st = TrlMctsStats()
        _check(self._L, self._L.trl_mcts_get_stats(self._h, C.byref(st)), "trl_mcts_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_}

    def mcts_leaf_states(self):
        states = np.empty((self.n_games, STATE_BYTES), dtype=np.uint8)
        info = np.empty((self.n_games, 4), dtype=np.int32)
        _check(self._L, self._L.trl_mcts_leaf_states(self._h, states.ctypes.data, info.ctypes.data), "trl_mcts_leaf_states")
        return states, info

    def mcts_root_children(self):
        n = self.n_games
        counts = np.empty(n, dtype=np.int32)
        moves = np.empty((n, MAX_LEGAL), dtype=np.uint16)
        visits = np.empty((n, MAX_LEGAL), dtype=np.int32)
        q = np.empty((n, MAX_LEGAL), dtype=np.float64)
        p = np.empty((n, MAX_LEGAL), dtype=np.float64)
        _check(self._L, self._L.trl_mcts_root_children(self._h, counts.ctypes.data, moves.ctypes.data, visits.ctypes.data,
                                                       q.ctypes.data, p.ctypes.data), "trl_mcts_root_children")
        return counts, moves, visits, q, p


def encode_move(plane, col, row):
    """reference move tuple (plane, board col, row) -> uint16"""
    return (int(plane) << 10) | (int(row) << 4) | (int(col) + 2)


def decode_move(m):
    m = int(m)
    return (m >> 10) & 31, (m & 15) - 2, (m >> 4) & 63
