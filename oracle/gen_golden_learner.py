#!/usr/bin/env python3
"""Golden fixture for the learner: runs the reference's train_network_pytorch
(ai.py:1139-1220) for one epoch on 24 records of a reference self-play game and stores the logged
losses and the updated weights (the 6 M-parameter policy head only as sampled rows).
TEST INFRASTRUCTURE ONLY; build container only.   python oracle/gen_golden_learner.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
import ref_harness as rh  # noqa: E402
import net_fixture  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
ROWS_SAMPLED = np.arange(0, 11583, 181)


def records_from_fixture(n):
    from tetris_rl_amd import records
    d = np.load(os.path.join(OUT, "selfplay.npz"))
    c_off = np.concatenate([[0], np.cumsum(d["ply_child_counts"])])
    data = []
    for k in range(int(d["game_plies"][0])):
        if d["ply_save"][k]:
            recs = records.records_for_move(d["ply_state"][k], d["ply_child_moves"][c_off[k]:c_off[k + 1]],
                                            d["ply_child_visits"][c_off[k]:c_off[k + 1]])
            for r in recs:
                r.insert(-1, 1 if (len(data) // 4) % 2 == 0 else 0)      # z
                data.append(r)
        if len(data) >= n:
            break
    return data[:n]


def main():
    import torch
    mods = rh.import_reference(with_ai=True)
    ai = mods["ai"]
    from architectures import AlphaSame, AlphaSameConfig, AuxBaseResNet, AuxBaseResNetConfig
    data = records_from_fixture(24)
    out = {}
    variants = (("alpha", AlphaSame, AlphaSameConfig(blocks=2, filters=8, dropout=0.0), 521),
                ("aux", AuxBaseResNet, AuxBaseResNetConfig(blocks=2, filters=8, opp_hidden=16, own_kernels=2, value_hidden=8, aux_hidden=8), 853))
    for tag, cls, mc, head_in in variants:
        torch.manual_seed(3)
        net = cls(mc)
        with torch.no_grad():
            w, b = net_fixture.policy_head(in_features=head_in)
            net.policy_head.weight.copy_(torch.from_numpy(w)); net.policy_head.bias.copy_(torch.from_numpy(b))
        for k, v in net.state_dict().items():
            if not k.startswith("policy_head."):
                out["%s/init/%s" % (tag, k)] = v.numpy().copy()
        cfg = ai.Config(visual=False, model="pytorch", model_config=mc, batch_size=8, shuffle=False, epochs=1,
                        learning_rate=0.001, weight_decay=0.01)
        log = ai.logs_path / "training_log.jsonl"
        if log.exists():
            log.unlink()
        ai.train_network_pytorch(cfg, net, [list(r) for r in data], data_number=0)
        import json
        entry = json.loads(open(log).read().strip().splitlines()[-1])
        out["%s/losses" % tag] = np.array([entry["loss"], entry["value_loss"], entry["policy_loss"], entry.get("aux_loss", 0.0)])
        for k, v in net.state_dict().items():
            if k == "policy_head.weight":
                out["%s/after/%s" % (tag, k)] = v.numpy()[ROWS_SAMPLED].copy()
            elif k == "policy_head.bias":
                out["%s/after/%s" % (tag, k)] = v.numpy()[ROWS_SAMPLED].copy()
            else:
                out["%s/after/%s" % (tag, k)] = v.numpy().copy()
        print(tag, entry)
    np.savez_compressed(os.path.join(OUT, "learner.npz"), **out)


if __name__ == "__main__":
    main()
