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

# every symbol include/tetris_engine.h declares (tests/test_abi.py checks the header against this)
ABI_SYMBOLS = [
    "trl_engine_create", "trl_engine_destroy", "trl_set_stream", "trl_last_error", "trl_version",
    "trl_env_reset", "trl_env_get_state", "trl_env_set_state", "trl_env_state_dev",
    "trl_env_legal_moves", "trl_env_legal_dev", "trl_env_step", "trl_env_step_dev",
    "trl_env_rollout_random", "trl_last_kernel_ms",
]


class TrlConfig(C.Structure):
    _fields_ = [("n_games", C.c_int32), ("ruleset", C.c_int32), ("device", C.c_int32), ("reserved", C.c_int32),
                ("seed", C.c_uint64), ("stream", C.c_void_p)]


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen libtrl_hip.so (building it first if the sources are newer) and set prototypes."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or _build.LIB_PATH
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
    L.trl_env_legal_dev.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    L.trl_env_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.trl_env_step_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.trl_env_rollout_random.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    L.trl_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    _lib = L
    return L


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

    def get_state(self):
        out = np.empty((self.n_games, STATE_BYTES), dtype=np.uint8)
        _check(self._L, self._L.trl_env_get_state(self._h, out.ctypes.data), "trl_env_get_state")
        return out

    def set_state(self, states):
        states = np.ascontiguousarray(states, dtype=np.uint8)
        if states.shape != (self.n_games, STATE_BYTES):
            raise ValueError("states must have shape (n_games, %d)" % STATE_BYTES)
        _check(self._L, self._L.trl_env_set_state(self._h, states.ctypes.data), "trl_env_set_state")

    def legal_moves(self):
        counts = np.empty(self.n_games, dtype=np.uint16)
        moves = np.empty((self.n_games, MAX_LEGAL), dtype=np.uint16)
        _check(self._L, self._L.trl_env_legal_moves(self._h, counts.ctypes.data, moves.ctypes.data), "trl_env_legal_moves")
        return counts, moves

    def step(self, moves, add_bag=True, check_legal=True):
        moves = np.ascontiguousarray(moves, dtype=np.uint16)
        if moves.shape != (self.n_games,):
            raise ValueError("moves must have shape (n_games,)")
        _check(self._L, self._L.trl_env_step(self._h, moves.ctypes.data, int(bool(add_bag)), int(bool(check_legal))),
               "trl_env_step")

    def rollout_random(self, n_plies, max_plies=2000, sync=True):
        steps = C.c_uint64(0)
        _check(self._L, self._L.trl_env_rollout_random(self._h, int(n_plies), int(max_plies),
                                                        C.byref(steps) if sync else None), "trl_env_rollout_random")
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


def encode_move(plane, col, row):
    """reference move tuple (plane, board col, row) -> uint16"""
    return (int(plane) << 10) | (int(row) << 4) | (int(col) + 2)


def decode_move(m):
    m = int(m)
    return (m >> 10) & 31, (m & 15) - 2, (m >> 4) & 63
