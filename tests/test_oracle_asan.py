"""Host-side sanitizer run of the oracle (AddressSanitizer + UBSan build of oracle/*.c): a rollout,
legal-move enumeration, a search and a whole stub-evaluator game must run clean.  CPU only; GPU
sanitizers are not available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import ctypes as C, sys
L = C.CDLL(sys.argv[1])
L.to_sizeof_game.restype = C.c_size_t
n = 24
buf = C.create_string_buffer(L.to_sizeof_game() * n)
L.to_batch_setup(buf, n, 2, C.c_uint64(20260630), None)
L.to_batch_rollout_random.restype = C.c_uint64
steps = L.to_batch_rollout_random(buf, n, 300, 2000)
assert steps > 0.9 * n * 300
moves = (C.c_uint16 * 512)()
L.to_legal_moves.restype = C.c_int
assert L.to_legal_moves(buf, moves) > 0
class Cfg(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("cpuct", C.c_double), ("dpuct", C.c_double), ("fpu_reduction", C.c_int),
                ("fpu_value", C.c_double), ("use_root_softmax", C.c_int), ("root_softmax_temp", C.c_double),
                ("temperature", C.c_double), ("training", C.c_int), ("use_playout_cap", C.c_int),
                ("playout_cap_chance", C.c_double), ("playout_cap_mult", C.c_double), ("use_dirichlet", C.c_int),
                ("dir_alpha", C.c_double), ("dir_s", C.c_double), ("dir_expl", C.c_double), ("use_dir_s", C.c_int),
                ("use_forced", C.c_int), ("c_forced", C.c_double)]
cfg = Cfg(24, 0.75, 1.0, 1, 0.1, 1, 1.1, 0.1, 1, 1, 0.25, 5.0, 1, 0.1, 25.0, 0.25, 1, 1, 1.0)
g = C.create_string_buffer(L.to_sizeof_game())
L.to_game_setup(g, 2, C.c_uint64(7), 3)
ml = 4096
st = C.create_string_buffer(320 * ml); mv = (C.c_uint16 * ml)(); sv = (C.c_int * ml)(); nc = (C.c_int * ml)()
cm = (C.c_uint16 * (ml * 160))(); cv = (C.c_int * (ml * 160))(); w = C.c_int()
plies = L.to_play_game_stub(g, C.byref(cfg), ml, st, mv, sv, nc, cm, cv, ml * 160, C.byref(w))
assert plies > 10
print("asan-ok", steps, plies)
"""


def test_oracle_runs_clean_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_build/liboracle_asan.so"])
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    out = subprocess.run([sys.executable, "-c", SCRIPT, lib], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "asan-ok" in out.stdout, out.stdout + out.stderr
