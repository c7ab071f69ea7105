"""Live cross-check of the oracle against the Python reference itself (build container only: skipped where
/root/reference is absent, e.g. on the GPU box).  The committed fixtures under tests/golden/ were produced
the same way (oracle/gen_golden.py); this test plays fresh games on other streams."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402
import oracle_py as op  # noqa: E402

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")


@pytest.mark.parametrize("ruleset,stream", [("s2", 7001), ("s1", 7002)])
def test_fresh_uniform_game_matches_reference(ruleset, stream):
    import gen_golden as gg
    seed = gg.SEED
    states, moves, legal = gg.play(ruleset, stream, "uniform", 80)
    g = op.OracleGame(ruleset=1 if ruleset == "s1" else 2, seed=seed, stream=stream)
    for k in range(len(moves) + 1):
        a, b = g.export(), np.frombuffer(states[k], dtype=np.uint8).copy()
        a[304:308] = 0; b[304:308] = 0
        assert np.array_equal(a, b), "ply %d" % k
        assert g.legal_moves().tolist() == legal[k], "legal moves at ply %d" % k
        if k < len(moves):
            g.make_move(moves[k])
