#!/usr/bin/env python3
"""Golden fixture for BASELINE config C5's net at its full depth: the reference's AlphaSame(blocks=20, filters=64)
(architectures.py:60-142; 40 convolutions of K = 576) on the eight positions of tests/golden/net.npz.  All weights
come from net_fixture.formula_state_dict on both sides (7.5 M parameters: nothing but the outputs is stored).
Also AlphaSame(10, 16) with formula weights and use_tanh (value range -1..1, architectures.py:116).

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_net_deep.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import net_fixture  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
TAGS = (("b20f64", 20, 64, False, 2064), ("b10f16tanh", 10, 16, True, 1016))


def main():
    import torch
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSame, AlphaSameConfig
    import gen_golden as gg

    states = np.load(os.path.join(OUT, "net.npz"))["states"]
    out = {}
    for tag, blocks, filters, use_tanh, salt in TAGS:
        net = AlphaSame(AlphaSameConfig(blocks=blocks, filters=filters), use_tanh=use_tanh)
        sd = net.state_dict()
        new = net_fixture.formula_state_dict({k: tuple(v.shape) for k, v in sd.items()}, salt=salt)
        net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in new.items()})
        net.eval()
        out["%s/keys" % tag] = np.array(list(sd.keys()))
        values, logits, v64, l64 = [], [], [], []
        net64 = AlphaSame(AlphaSameConfig(blocks=blocks, filters=filters), use_tanh=use_tanh).double()
        net64.load_state_dict({k: torch.from_numpy(np.asarray(v)).double() if v.dtype != np.int64 else torch.from_numpy(v)
                               for k, v in new.items()})
        net64.eval()
        for st in states:
            game, rng, ply = gg.game_from_state(bytes(st))
            X = []
            for feature in ai.game_to_X(game):        # evaluate_pytorch's tensor construction (ai.py:1296-1313)
                t = torch.tensor(feature)
                if t.size() == (40, 10):
                    t = t.unsqueeze(dim=0).type(torch.float)
                X.append(t.unsqueeze(dim=0))
            with torch.no_grad():
                v, p = net.forward(*X)
                vd, pd = net64.forward(*[x.double() for x in X])
            values.append(float(v.item()))
            logits.append(p.reshape(-1).numpy().copy())
            v64.append(float(vd.item())); l64.append(pd.reshape(-1).numpy().copy())
        out["%s/values" % tag] = np.array(values, dtype=np.float32)
        out["%s/logits" % tag] = np.stack(logits)
        # how far the reference's own fp32 run is from exact arithmetic on the same weights (the floor of any fp32 parity)
        out["%s/ref_fp32_vs_fp64_max" % tag] = np.array([np.abs(np.stack(logits) - np.stack(l64)).max(),
                                                         np.abs(np.array(values) - np.array(v64)).max()])
        print(tag, "values", np.round(values, 4), "logit range", float(np.min(logits)), float(np.max(logits)),
              "fp32 vs fp64", out["%s/ref_fp32_vs_fp64_max" % tag])
    out["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "net_deep.npz"), **out)


if __name__ == "__main__":
    main()
