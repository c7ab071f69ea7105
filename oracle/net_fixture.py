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


def formula_state_dict(shapes, salt=0, branch_gain=0.5):
    """Every tensor of a net by formula (nothing to commit, any depth / width): `shapes` is the ordered
    {key: shape} of a state_dict.  For the k-th key and flat index i, u = u24(mix64(((k + 1) << 40) + i + salt)) / 2^24;
    batch-norm weight 0.8 + 0.4 u, bias / Linear bias (u - 0.5) / 10, running_mean (u - 0.5) / 5, running_var 0.5 + u;
    conv / Linear weights uniform with variance 1 / fan_in (the second conv of a residual block scaled by `branch_gain`
    so that the residual stream of a 20-block tower stays O(1)); num_batches_tracked = 0."""
    out = {}
    for k, (key, shape) in enumerate(shapes.items()):
        n = int(np.prod(shape)) if len(shape) else 1
        idx = np.arange(n, dtype=np.uint64) + (np.uint64(k + 1) << np.uint64(40)) + np.uint64(salt)
        u = (_mix64(idx) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = np.zeros(shape, dtype=np.int64)
            continue
        if leaf == "running_var":
            v = np.float32(0.5) + u
        elif leaf == "running_mean":
            v = (u - np.float32(0.5)) * np.float32(0.2)
        elif leaf == "bias":
            v = (u - np.float32(0.5)) * np.float32(0.1)
        elif len(shape) == 1:                                     # batch-norm weight
            v = np.float32(0.8) + np.float32(0.4) * u
        else:
            fan_in = int(np.prod(shape[1:]))
            gain = branch_gain if key.endswith("conv_block2.3.weight") or key.endswith("conv2.weight") else 1.0
            v = (u - np.float32(0.5)) * np.float32(2.0 * gain * np.sqrt(3.0 / fan_in))
        out[key] = v.astype(np.float32).reshape(shape)
    return out
