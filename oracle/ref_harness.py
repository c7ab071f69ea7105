"""Import recipe for the Python reference (build container only) + the engine RNG in Python.

TEST INFRASTRUCTURE ONLY (see oracle/tetris_oracle.h).  The reference lives at
/root/reference, is never copied and never travels to the GPU box; this module is used by
oracle/gen_golden.py to produce the fixtures under tests/golden/ and by
tests/test_oracle_vs_reference.py, which skips itself when /root/reference is absent.

Recipe: stub the UI/TF modules the reference imports at module level
(const.py:1 pygame; ai.py:24-29 ujson/tensorflow/keras), put /root/reference on sys.path,
run from a cwd whose parent has a Storage/ directory (ai.py:57-59).
"""
import json
import os
import sys
from unittest.mock import MagicMock

REFERENCE_DIR = os.environ.get("TRL_REFERENCE_DIR", "/root/reference")

MASK64 = (1 << 64) - 1
MINOS = "ZLOSIJT"
RNG_BAG, RNG_GARBAGE, RNG_POLICY = 0, 1, 2


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_DIR, "move_generation.py"))


def import_reference(with_ai=False):
    """Returns a dict of the reference modules needed for the env path (and ai.py if asked)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_DIR)
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    stubs = ["pygame"]
    if with_ai:
        stubs += ["tensorflow", "tensorflow.python", "tensorflow.python.ops", "keras", "absl",
                  "absl.logging", "imageio", "imageio.v3", "skopt", "matplotlib", "matplotlib.pyplot",
                  "sklearn", "sklearn.model_selection"]
        sys.modules.setdefault("ujson", json)
    import importlib
    import importlib.machinery
    import types

    class _Stub(types.ModuleType):             # a module object (find_spec works on it) whose attributes are mocks
        def __getattr__(self, key):
            if key.startswith("__"):
                raise AttributeError(key)
            m = MagicMock()
            setattr(self, key, m)
            return m

    for name in stubs:
        if name in sys.modules:
            continue
        try:                                   # stub only what is really absent (sklearn, matplotlib exist here)
            importlib.import_module(name)
        except Exception:
            mod = _Stub(name)
            mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
            mod.__path__ = []
            sys.modules[name] = mod
    if REFERENCE_DIR not in sys.path:
        sys.path.insert(0, REFERENCE_DIR)
    mods = {}
    if with_ai:
        run_dir = "/tmp/trl_oracle/run"
        os.makedirs("/tmp/trl_oracle/Storage", exist_ok=True)
        os.makedirs(run_dir, exist_ok=True)
        old = os.getcwd()
        os.chdir(run_dir)
        try:
            import ai  # noqa
        finally:
            os.chdir(old)
        mods["ai"] = ai
        import architectures
        mods["architectures"] = architectures
    import const, game, player, piece, piece_queue, move_generation, stats, board
    mods.update(const=const, game=game, player=player, piece=piece, piece_queue=piece_queue,
                move_generation=move_generation, stats=stats, board=board)
    return mods


# ---- the engine's counter-based RNG (definition: include/tetris_engine.h) -----------------
def mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def rng_u32(seed, stream, purpose, ctr):
    k = mix64((seed ^ ((stream * 0xD1342543DE82EF95) & MASK64)) & MASK64)
    return mix64(k ^ ((purpose << 56) & MASK64) ^ ctr) >> 32


def rng_below(seed, stream, purpose, ctr, n):
    return (rng_u32(seed, stream, purpose, ctr) * n) >> 32


class EngineRNG:
    """Stands in for the `random` module inside piece_queue.py and player.py.

    piece_queue.py:19 calls random.shuffle(list("ZLOSIJT")); player.py:185 calls
    random.randint(0, 9).  Both are served from the engine's generator so that a reference
    game and an engine game with the same (seed, stream) see the same bags and columns.
    """

    def __init__(self, seed, stream):
        self.seed, self.stream = seed, stream
        self.bag_ctr = 0
        self.garb_ctr = 0
        self.pol_ctr = 0

    def shuffle(self, lst):
        assert len(lst) == 7
        k = self.bag_ctr
        self.bag_ctr += 1
        for i in range(6, 0, -1):
            j = rng_below(self.seed, self.stream, RNG_BAG, 8 * k + i, i + 1)
            lst[i], lst[j] = lst[j], lst[i]

    def randint(self, a, b):
        assert (a, b) == (0, 9)
        c = rng_below(self.seed, self.stream, RNG_GARBAGE, self.garb_ctr, 10)
        self.garb_ctr += 1
        return c

    def policy_index(self, n):
        i = rng_below(self.seed, self.stream, RNG_POLICY, self.pol_ctr, n)
        self.pol_ctr += 1
        return i

    def snapshot(self):
        return (self.bag_ctr, self.garb_ctr, self.pol_ctr)

    def restore(self, s):
        self.bag_ctr, self.garb_ctr, self.pol_ctr = s


def install_rng(mods, rng):
    """piece_queue.random / player.random -> rng (no reference file is modified)."""
    mods["piece_queue"].random = rng
    mods["player"].random = rng


# ---- canonical state (trl_game_state, 320 bytes) from reference objects ---------------------
import struct  # noqa: E402

PLAYER_FMT = "<40H4QQiiihhBBBBBB2x"
GAME_TAIL_FMT = "<IIIIIIBBB5x"
assert struct.calcsize(PLAYER_FMT) == 144 and struct.calcsize(GAME_TAIL_FMT) == 32
NO_PIECE = 7


def pack_player(pl):
    rows = []
    for r in range(40):
        bits = 0
        for c in range(10):
            if pl.board.grid[r][c] != 0:
                bits |= 1 << c
        rows.append(bits)
    garb = [0, 0, 0, 0]
    overflow = False
    gl = len(pl.garbage_to_receive)
    if gl > 64:
        overflow, gl = True, 64
    for k in range(gl):
        garb[k >> 4] |= pl.garbage_to_receive[k] << (4 * (k & 15))
    ql = len(pl.queue.pieces)
    if ql > 16:
        overflow, ql = True, 16
    q = 0
    for k in range(ql):
        q |= MINOS.index(pl.queue.pieces[k]) << (4 * k)
    piece = MINOS.index(pl.piece.type) if pl.piece is not None else NO_PIECE
    held = MINOS.index(pl.held_piece) if pl.held_piece is not None else NO_PIECE
    st = pl.stats
    data = struct.pack(PLAYER_FMT, *rows, *garb, q, st.pieces, st.lines_sent, st.lines_cleared,
                       st.b2b, st.combo, st.b2b_level, piece, held, ql, gl, 1 if pl.game_over else 0)
    return data, overflow


def pack_game(g, rng, ply, ruleset, episode=0):
    p0, o0 = pack_player(g.players[0])
    p1, o1 = pack_player(g.players[1])
    flags = 1 if (o0 or o1) else 0
    rs = 1 if ruleset == "s1" else 2
    tail = struct.pack(GAME_TAIL_FMT, rng.stream, rng.bag_ctr, rng.garb_ctr, ply, rng.pol_ctr, episode,
                       g.turn, flags, rs)
    return p0 + p1 + tail


def encode_move(plane, col, row):
    """(plane, col, row) of the reference (board col) -> uint16 of include/tetris_engine.h"""
    return (plane << 10) | (row << 4) | (col + 2)


def mask_to_moves(mask):
    import numpy as np
    idx = np.argwhere(mask)
    return [(int(p) << 10) | (int(r) << 4) | int(c) for p, r, c in idx]
