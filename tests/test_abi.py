"""CPU-side checks of the drop-in boundary: the HIP library builds for gfx950, loads, exports
every symbol include/tetris_engine.h declares, and refuses to run without a GPU (no fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import tetris_rl_amd as trl
from tetris_rl_amd import engine as eng

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "tetris_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(trl_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    path = trl.build()
    assert os.path.isfile(path)
    lib = ctypes.CDLL(path)
    declared = _declared_functions()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libtrl_hip.so does not export %s" % name
    assert sorted(eng.ABI_SYMBOLS) == declared


def test_struct_layout_matches_header():
    # 144-byte player, 320-byte game (include/tetris_engine.h); offsets used by the tests / harness
    text = open(os.path.join(ROOT, "include", "tetris_engine.h")).read()
    assert "144 bytes" in text and "320 bytes" in text
    assert eng.STATE_BYTES == 320 and eng.MAX_LEGAL == 512


def test_move_encoding_roundtrip():
    for plane, col, row in [(0, 0, 38), (26, 8, 0), (15, -1, 20), (5, -2, 17)]:
        m = trl.encode_move(plane, col, row)
        assert trl.decode_move(m) == (plane, col, row)
    # ascending code == argwhere order (plane, row, col)
    codes = [trl.encode_move(p, c, r) for p in range(27) for r in range(39) for c in range(-2, 9)]
    assert codes == sorted(codes)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(trl.EngineError):
        trl.Engine(4)


def test_bad_arguments():
    with pytest.raises(ValueError):
        trl.Engine(4, ruleset="s3")
