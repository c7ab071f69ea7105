"""Learner (train_network_pytorch, ai.py:1139-1220) against the reference: one epoch over 24 records of a
reference self-play game; logged losses and every updated tensor (policy head: sampled rows) within
1e-5 of the reference's.  CPU (PyTorch); the learner is plain PyTorch on either device."""
import os
import sys

import numpy as np
import pytest
import torch

import net_fixture
import tetris_rl_amd as trl
from tetris_rl_amd import learner, nets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
ROWS_SAMPLED = np.arange(0, 11583, 181)


def _data(golden_dir, n=24):
    import gen_golden_learner as g          # the same record construction as the fixture (no reference import at module level)
    return g.records_from_fixture(n)


@pytest.mark.parametrize("tag", ["alpha", "aux"])
def test_one_epoch_matches_reference(golden_dir, tag, tmp_path, monkeypatch):
    monkeypatch.setenv("TRL_STORAGE", str(tmp_path))
    d = np.load(os.path.join(golden_dir, "learner.npz"))
    if tag == "alpha":
        mc = nets.AlphaSameConfig(blocks=2, filters=8, dropout=0.0)
        net = nets.AlphaSame(mc)
    else:
        mc = nets.AuxBaseResNetConfig(blocks=2, filters=8, opp_hidden=16, own_kernels=2, value_hidden=8, aux_hidden=8)
        net = nets.AuxBaseResNet(mc)
    head_in = net.policy_head.weight.shape[1]
    w, b = net_fixture.policy_head(in_features=head_in)
    sd = {k: (torch.from_numpy(w) if k == "policy_head.weight" else torch.from_numpy(b) if k == "policy_head.bias"
              else torch.from_numpy(d["%s/init/%s" % (tag, k)])) for k in net.state_dict().keys()}
    net.load_state_dict(sd)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, batch_size=8, shuffle=False, epochs=1,
                     learning_rate=0.001, weight_decay=0.01)
    entry = learner.train_network_pytorch(cfg, net, _data(golden_dir), data_number=0)
    want = d["%s/losses" % tag]
    got = [entry["loss"], entry["value_loss"], entry["policy_loss"], entry.get("aux_loss", 0.0)]
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6), (got, want)
    after = net.state_dict()
    for k, v in after.items():
        ref = d["%s/after/%s" % (tag, k)]
        mine = v.numpy()[ROWS_SAMPLED] if k.startswith("policy_head.") else v.numpy()
        assert np.allclose(mine, ref, rtol=1e-4, atol=1e-5), k
    assert os.path.isfile(tmp_path / "logs" / "training_log.jsonl")
    # the inference path sees the new weights (folded tensors are rebuilt)
    assert net._folded is None
