"""Shared by oracle/gen_golden_net.py and the net parity tests (TEST INFRASTRUCTURE ONLY).

The AlphaSame policy head is 521 x 11583 = 6.03 M parameters (24 MB) -- too large to commit --
so the fixture fixes it by formula: w[i, j] = (u24(i * 521 + j) / 2^24 - 0.5) * 0.1 with u24 the
top 24 bits of the engine's mix64; bias likewise from index 2^32 + i.  Everything else of the
net (trunk, dense heads, batch-norm statistics) is stored in tests/golden/net.npz.
"""
import numpy as np

GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    with np.errstate(over="ignore"):
        z = z + GOLD
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u(idx):
    k = _mix64(idx.astype(np.uint64))
    return ((k >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0) - np.float32(0.5)) * np.float32(0.1)


def policy_head(out_features=11583, in_features=521):
    w = _u(np.arange(out_features * in_features, dtype=np.uint64)).reshape(out_features, in_features)
    b = _u(np.arange(out_features, dtype=np.uint64) + np.uint64(1 << 32))
    return w, b
