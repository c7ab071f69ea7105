#!/usr/bin/env python3
"""Golden fixture for the policy/value net: runs the reference's
AlphaSame (architectures.py:60-142) on reference features (game_to_X, ai.py:1399-1413) of eight
positions and stores inputs, weights (except the formula-defined policy head) and outputs.

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_net.py
Parity for this row is anchored on the reference's own module run here (torch %s); the reference
has no test that pins net outputs.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import net_fixture  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    import torch
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSame, AlphaSameConfig
    import gen_golden as gg

    torch.manual_seed(0)
    out = {}
    for tag, blocks, filters in (("b10f16", 10, 16), ("b2f8", 2, 8)):
        net = AlphaSame(AlphaSameConfig(blocks=blocks, filters=filters))
        g = torch.Generator().manual_seed(1234)
        with torch.no_grad():
            for name, mod in net.named_modules():
                if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                    mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                    mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.4 + 0.8)
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.05)
            w, b = net_fixture.policy_head()
            net.policy_head.weight.copy_(torch.from_numpy(w))
            net.policy_head.bias.copy_(torch.from_numpy(b))
        net.eval()
        sd = net.state_dict()
        for k, v in sd.items():
            if not k.startswith("policy_head."):
                out["%s/%s" % (tag, k)] = v.numpy()
        out["%s/keys" % tag] = np.array(list(sd.keys()))

        tr = np.load(os.path.join(OUT, "env_traces.npz"))
        meta = tr["meta"]
        picks = [int(meta[gi, 4]) + k for gi, k in ((24, 0), (24, 40), (27, 90), (31, 61), (64, 33), (67, 120), (70, 77), (75, 50))]
        states = tr["states"][picks]
        values, logits, feats = [], [], []
        for st in states:
            game, rng, ply = gg.game_from_state(bytes(st))
            data = ai.game_to_X(game)
            X = []
            for feature in data:                      # evaluate_pytorch's tensor construction (ai.py:1296-1313)
                t = torch.tensor(feature)
                if t.size() == (40, 10):
                    t = t.unsqueeze(dim=0).type(torch.float)
                X.append(t.unsqueeze(dim=0))
            with torch.no_grad():
                v, p = net.forward(*X)
            values.append(float(v.item()))
            logits.append(p.reshape(-1).numpy().copy())
            feats.append(np.concatenate([np.asarray(f, dtype=np.float32).reshape(-1) for f in data]))
        out["%s/values" % tag] = np.array(values, dtype=np.float32)
        out["%s/logits" % tag] = np.stack(logits)
        out["states"] = states
        out["ref_features"] = np.stack(feats)       # 400+49+3+400+49+3+1 = 905 numbers in game_to_X order
        print(tag, "values", np.round(values, 4), "logit range", float(np.min(logits)), float(np.max(logits)))
    # ---- the reference's current default family: BaseResNet / AuxBaseResNet (architectures.py:150-353) ----
    from architectures import BaseResNet, BaseResNetConfig, AuxBaseResNet, AuxBaseResNetConfig
    small = dict(blocks=2, filters=8, opp_hidden=16, own_kernels=2, value_hidden=8)
    for tag, cls, mc in (("base", BaseResNet, BaseResNetConfig(**small)),
                         ("aux", AuxBaseResNet, AuxBaseResNetConfig(aux_hidden=8, **small))):
        torch.manual_seed(5)
        net = cls(mc)
        g = torch.Generator().manual_seed(77)
        head_in = mc.own_kernels * 400 + 49 + 3 + 1
        with torch.no_grad():
            for name, mod in net.named_modules():
                if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                    mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
                    mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
                    mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 0.4 + 0.8)
                    mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.05)
            w, b = net_fixture.policy_head(in_features=head_in)
            net.policy_head.weight.copy_(torch.from_numpy(w))
            net.policy_head.bias.copy_(torch.from_numpy(b))
        net.eval()
        sd = net.state_dict()
        for k, v in sd.items():
            if not k.startswith("policy_head."):
                out["%s/%s" % (tag, k)] = v.numpy()
        out["%s/keys" % tag] = np.array(list(sd.keys()))
        values, logits, auxs = [], [], []
        for st in states:
            game, rng, ply = gg.game_from_state(bytes(st))
            X = []
            for feature in ai.game_to_X(game):
                t = torch.tensor(feature)
                if t.size() == (40, 10):
                    t = t.unsqueeze(dim=0).type(torch.float)
                X.append(t.unsqueeze(dim=0))
            with torch.no_grad():
                o = net.forward(*X)
            values.append(float(o[0].item())); logits.append(o[1].reshape(-1).numpy().copy())
            if len(o) == 3:
                auxs.append(o[2].reshape(-1).numpy().copy())
        out["%s/values" % tag] = np.array(values, dtype=np.float32)
        out["%s/logits" % tag] = np.stack(logits)
        if auxs:
            out["%s/aux" % tag] = np.stack(auxs)
        print(tag, "values", np.round(values, 4))
    # key lists of the default-size nets (names and shapes only)
    for tag, net in (("base_default", BaseResNet(BaseResNetConfig())), ("aux_default", AuxBaseResNet(AuxBaseResNetConfig()))):
        sd = net.state_dict()
        out["%s/keys" % tag] = np.array(list(sd.keys()))
        out["%s/shapes" % tag] = np.array([str(tuple(v.shape)) for v in sd.values()])
    out["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "net.npz"), **out)


if __name__ == "__main__":
    main()
