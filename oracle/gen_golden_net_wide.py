#!/usr/bin/env python3
"""Golden fixture for the wide AlphaSame towers (32 / 64 filters; BASELINE config C5 is 20 blocks x 64 filters):
the reference's AlphaSame (architectures.py:60-142) run on the eight positions of tests/golden/net.npz.
Same recipe as oracle/gen_golden_net.py (weights stored except the formula-defined policy head), fewer blocks
than C5 so that the file stays small -- the kernel's block loop does not depend on the count.

TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_net_wide.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness as rh  # noqa: E402
import net_fixture  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
TAGS = (("b3f64", 3, 64), ("b2f32", 2, 32))


def main():
    import torch
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSame, AlphaSameConfig
    import gen_golden as gg

    states = np.load(os.path.join(OUT, "net.npz"))["states"]
    out = {"states": states}
    for tag, blocks, filters in TAGS:
        torch.manual_seed(blocks * 1000 + filters)
        net = AlphaSame(AlphaSameConfig(blocks=blocks, filters=filters))
        g = torch.Generator().manual_seed(4321)
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
        values, logits = [], []
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
            values.append(float(v.item()))
            logits.append(p.reshape(-1).numpy().copy())
        out["%s/values" % tag] = np.array(values, dtype=np.float32)
        out["%s/logits" % tag] = np.stack(logits)
        print(tag, "values", np.round(values, 4), "logit range", float(np.min(logits)), float(np.max(logits)))
    # ---- the default family at a width the fused tower covers: BaseResNet / AuxBaseResNet, 32 filters ----
    from architectures import BaseResNet, BaseResNetConfig, AuxBaseResNet, AuxBaseResNetConfig
    small = dict(blocks=3, filters=32, opp_hidden=16, own_kernels=4, value_hidden=8)
    for tag, cls, mc in (("base32", BaseResNet, BaseResNetConfig(**small)),
                         ("aux32", AuxBaseResNet, AuxBaseResNetConfig(aux_hidden=8, **small))):
        torch.manual_seed(11)
        net = cls(mc)
        g = torch.Generator().manual_seed(78)
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
    out["torch_version"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(OUT, "net_wide.npz"), **out)


if __name__ == "__main__":
    main()
