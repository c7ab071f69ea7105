"""AlphaSame inference module against the reference's module (fixture oracle/gen_golden_net.py):
same state_dict keys, value and all 11 583 policy logits within 1e-5 (fp32), for two
architectures.  CPU here; tests/test_hip_net.py repeats it on the GPU with the packed features
the search kernel writes."""
import os

import numpy as np
import pytest
import torch

import net_fixture
from tetris_rl_amd import nets, records

TOL = 1e-5


def load_net(d, tag, blocks, filters, device="cpu"):
    net = nets.AlphaSame(nets.AlphaSameConfig(blocks=blocks, filters=filters))
    keys = [str(k) for k in d["%s/keys" % tag]]
    assert list(net.state_dict().keys()) == keys, "state_dict keys differ from the reference's"
    sd = {}
    w, b = net_fixture.policy_head()
    for k in keys:
        if k == "policy_head.weight":
            sd[k] = torch.from_numpy(w)
        elif k == "policy_head.bias":
            sd[k] = torch.from_numpy(b)
        else:
            sd[k] = torch.from_numpy(d["%s/%s" % (tag, k)])
    net.load_state_dict(sd)
    return net.to(device).prepare()


def packed_features(states):
    """grids [2, n, 400], vec [n, 105] from canonical states via the host-side game_to_X"""
    n = len(states)
    grids = np.zeros((2, n, 400), dtype=np.float32)
    vec = np.zeros((n, 105), dtype=np.float32)
    for i, st in enumerate(states):
        x = records.game_to_x(st)
        grids[0, i], grids[1, i] = x[0].reshape(-1), x[5].reshape(-1)
        vec[i] = np.concatenate([x[1].reshape(-1), [x[2], x[3], x[4]], x[6].reshape(-1), [x[7], x[8], x[9]], [x[10]]])
    return grids, vec


@pytest.mark.parametrize("tag,blocks,filters", [("b10f16", 10, 16), ("b2f8", 2, 8)])
def test_logits_and_value_match_reference(golden_dir, tag, blocks, filters):
    d = np.load(os.path.join(golden_dir, "net.npz"))
    net = load_net(d, tag, blocks, filters)
    grids, vec = packed_features(d["states"])
    # host-side features == the reference's game_to_X output
    ref = d["ref_features"]
    mine = np.concatenate([grids[0], vec[:, :52], grids[1], vec[:, 52:]], axis=1)
    assert np.array_equal(mine, ref)
    v, logits = net.heads(torch.from_numpy(grids), torch.from_numpy(vec))
    v = torch.sigmoid(v).numpy()
    assert np.abs(v - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.numpy() - d["%s/logits" % tag]).max() < TOL
    # the reference's 11-tensor signature gives the same numbers
    n = len(d["states"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    out_v, out_p = net(t(grids[0]).reshape(n, 1, 40, 10), t(vec[:, :49]).reshape(n, 7, 7), t(vec[:, 49]), t(vec[:, 50]),
                       t(vec[:, 51]), t(grids[1]).reshape(n, 1, 40, 10), t(vec[:, 52:101]).reshape(n, 7, 7), t(vec[:, 101]),
                       t(vec[:, 102]), t(vec[:, 103]), t(vec[:, 104]))
    assert np.abs(out_p.numpy() - d["%s/logits" % tag]).max() < TOL
    assert np.abs(out_v.numpy().reshape(-1) - d["%s/values" % tag]).max() < TOL


@pytest.mark.parametrize("tag,blocks,filters", [("b3f64", 3, 64), ("b2f32", 2, 32)])
def test_wide_towers_match_reference(golden_dir, tag, blocks, filters):
    """32 / 64-filter AlphaSame (BASELINE config C5 is 20 x 64) against the reference's outputs"""
    d = np.load(os.path.join(golden_dir, "net_wide.npz"))
    net = load_net(d, tag, blocks, filters)
    grids, vec = packed_features(d["states"])
    v, logits = net.heads(torch.from_numpy(grids), torch.from_numpy(vec))
    assert np.abs(torch.sigmoid(v).numpy() - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.numpy() - d["%s/logits" % tag]).max() < TOL


def load_formula_net(blocks, filters, salt, use_tanh=False, device="cpu"):
    """AlphaSame with every weight from net_fixture.formula_state_dict (the recipe of oracle/gen_golden_net_deep.py)"""
    net = nets.AlphaSame(nets.AlphaSameConfig(blocks=blocks, filters=filters), use_tanh=use_tanh)
    sd = net_fixture.formula_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, salt=salt)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return net.to(device).prepare()


DEEP = {"b20f64": (20, 64, 2064, False), "b10f16tanh": (10, 16, 1016, True)}


@pytest.mark.parametrize("tag", sorted(DEEP))
def test_deep_and_tanh_nets_match_reference(golden_dir, tag):
    """BASELINE config C5's net at full depth (20 blocks x 64 filters = 40 convolutions of K = 576) and the tanh value
    head (architectures.py:116), against the reference's outputs on the net.npz positions"""
    blocks, filters, salt, use_tanh = DEEP[tag]
    d = np.load(os.path.join(golden_dir, "net_deep.npz"))
    net = load_formula_net(blocks, filters, salt, use_tanh)
    assert list(net.state_dict().keys()) == [str(k) for k in d["%s/keys" % tag]]
    grids, vec = packed_features(np.load(os.path.join(golden_dir, "net.npz"))["states"])
    v, logits = net.heads(torch.from_numpy(grids), torch.from_numpy(vec))
    v = (torch.tanh(v) if use_tanh else torch.sigmoid(v)).numpy()
    assert np.abs(v - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.numpy() - d["%s/logits" % tag]).max() < TOL
    if use_tanh:
        assert (d["%s/values" % tag] < 0).any()                 # the fixture does exercise the negative half of the range


NET64 = ("b10f16", "b10f16tanh", "b20f64", "b2f32")


def load_net64(golden_dir, tag, device="cpu"):
    """the nets of tests/golden/net64.npz (oracle/gen_golden_net64.py): stored weights or formula weights, as there"""
    if tag == "b10f16":
        return load_net(np.load(os.path.join(golden_dir, "net.npz")), "b10f16", 10, 16, device=device), False
    if tag == "b2f32":
        return load_net(np.load(os.path.join(golden_dir, "net_wide.npz")), "b2f32", 2, 32, device=device), False
    blocks, filters, salt, use_tanh = DEEP[tag]
    return load_formula_net(blocks, filters, salt, use_tanh, device=device), use_tanh


def check_net64(golden_dir, tag, net, use_tanh, device="cpu", tol=TOL):
    """value, 768 sampled logits, and the sum / L2 norm of all 11 583 logits of the 64 fixture positions (tall stacks,
    garbage-heavy and nearly empty boards: what the activations' magnitude depends on) against the reference's"""
    d = np.load(os.path.join(golden_dir, "net64.npz"))
    grids, vec = packed_features(d["states"])
    v, logits = net.heads(torch.from_numpy(grids).to(device), torch.from_numpy(vec).to(device))
    v = (torch.tanh(v) if use_tanh else torch.sigmoid(v)).cpu().numpy()
    lg = logits.cpu().numpy()
    err = np.abs(lg[:, d["sampled"]] - d["%s/sampled_logits" % tag]).max()
    assert np.abs(v - d["%s/values" % tag]).max() < tol
    assert err < tol, err
    # sums over 11 583 logits of magnitude ~1: per-logit errors of 1e-5 could add up to 0.1; they do not get near it
    assert np.abs(lg.astype(np.float64).sum(axis=1) - d["%s/logit_sum" % tag]).max() < 5e-3
    assert np.abs(np.sqrt((lg.astype(np.float64) ** 2).sum(axis=1)) - d["%s/logit_l2" % tag]).max() < 1e-4
    return err


@pytest.mark.parametrize("tag", NET64)
def test_64_positions_match_reference(golden_dir, tag):
    net, use_tanh = load_net64(golden_dir, tag)
    check_net64(golden_dir, tag, net, use_tanh)


def scaled_copy(net, what):
    """b10f16 with one block rescaled by a power of two -- the same function (and, powers of two being exact, the same
    fp32 results in the reference's own forward), but a different operand range for the split tower kernels:
      'weight': block 3's second batch norm x 1024, its second conv / 1024 -> folded first-conv weights up to ~85 (>= 31:
                the packer refuses the split layout, the fp32 kernels run);
      'activation': block 3's first batch norm x 2^17, its first conv / 2^17 -> relu(bn1(x)) reaches ~1e6 (>= 65 504: no
                f16 hi half), every folded weight stays small."""
    import copy
    m = copy.deepcopy(net)
    blk = m.res_blocks[3]
    with torch.no_grad():
        if what == "weight":
            blk.conv_block2["0"].weight.mul_(1024.0); blk.conv_block2["0"].bias.mul_(1024.0)
            blk.conv_block2["3"].weight.div_(1024.0)
        else:
            blk.conv_block1["0"].weight.mul_(131072.0); blk.conv_block1["0"].bias.mul_(131072.0)
            blk.conv_block1["2"].weight.div_(131072.0)
    m._folded = None
    return m


def test_reference_checkpoint_roundtrip(tmp_path, golden_dir):
    d = np.load(os.path.join(golden_dir, "net.npz"))
    net = load_net(d, "b2f8", 2, 8)
    path = os.path.join(tmp_path, "3.pt")
    torch.save(net.state_dict(), path)
    net2 = nets.load_reference_checkpoint(path, nets.AlphaSameConfig(blocks=2, filters=8))
    grids, vec = packed_features(d["states"])
    a = net.evaluate_packed(torch.from_numpy(grids), torch.from_numpy(vec))
    b = net2.evaluate_packed(torch.from_numpy(grids), torch.from_numpy(vec))
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert abs(float(a[1].sum(dim=1)[0]) - 1.0) < 1e-5


def _load_generic(d, tag, net):
    keys = [str(k) for k in d["%s/keys" % tag]]
    assert list(net.state_dict().keys()) == keys, "state_dict keys differ from the reference's"
    head_in = net.policy_head.weight.shape[1]
    w, b = net_fixture.policy_head(in_features=head_in)
    sd = {k: (torch.from_numpy(w) if k == "policy_head.weight" else torch.from_numpy(b) if k == "policy_head.bias"
              else torch.from_numpy(d["%s/%s" % (tag, k)])) for k in keys}
    net.load_state_dict(sd)
    return net.prepare()


@pytest.mark.parametrize("tag", ["base", "aux"])
def test_baseresnet_family_matches_reference(golden_dir, tag):
    """BaseResNet / AuxBaseResNet (the reference's current default family, architectures.py:150-353)"""
    d = np.load(os.path.join(golden_dir, "net.npz"))
    small = dict(blocks=2, filters=8, opp_hidden=16, own_kernels=2, value_hidden=8)
    net = (nets.BaseResNet(nets.BaseResNetConfig(**small)) if tag == "base"
           else nets.AuxBaseResNet(nets.AuxBaseResNetConfig(aux_hidden=8, **small)))
    net = _load_generic(d, tag, net)
    grids, vec = packed_features(d["states"])
    v, logits = net.heads(torch.from_numpy(grids), torch.from_numpy(vec))
    assert np.abs(torch.sigmoid(v).numpy() - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.numpy() - d["%s/logits" % tag]).max() < TOL
    if tag == "aux":
        n = len(d["states"])
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        out = net(t(grids[0]).reshape(n, 1, 40, 10), t(vec[:, :49]).reshape(n, 7, 7), t(vec[:, 49]), t(vec[:, 50]), t(vec[:, 51]),
                  t(grids[1]).reshape(n, 1, 40, 10), t(vec[:, 52:101]).reshape(n, 7, 7), t(vec[:, 101]), t(vec[:, 102]),
                  t(vec[:, 103]), t(vec[:, 104]))
        assert len(out) == 3 and np.abs(out[2].numpy() - d["aux/aux"]).max() < TOL


SMALL32 = dict(blocks=3, filters=32, opp_hidden=16, own_kernels=4, value_hidden=8)


def load_base32(golden_dir, tag, device="cpu"):
    d = np.load(os.path.join(golden_dir, "net_wide.npz"))
    net = (nets.BaseResNet(nets.BaseResNetConfig(**SMALL32)) if tag == "base32"
           else nets.AuxBaseResNet(nets.AuxBaseResNetConfig(aux_hidden=8, **SMALL32)))
    return d, _load_generic(d, tag, net).to(device).prepare()


@pytest.mark.parametrize("tag", ["base32", "aux32"])
def test_baseresnet_32_filters_matches_reference(golden_dir, tag):
    """the default family at a width the fused HIP tower covers (here: its PyTorch path on the CPU)"""
    d, net = load_base32(golden_dir, tag)
    grids, vec = packed_features(d["states"])
    v, logits = net.heads(torch.from_numpy(grids), torch.from_numpy(vec))
    assert np.abs(torch.sigmoid(v).numpy() - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.numpy() - d["%s/logits" % tag]).max() < TOL


def test_default_size_nets_have_the_reference_key_layout(golden_dir):
    d = np.load(os.path.join(golden_dir, "net.npz"))
    for tag, net in (("base_default", nets.BaseResNet()), ("aux_default", nets.AuxBaseResNet())):
        sd = net.state_dict()
        assert list(sd.keys()) == [str(k) for k in d["%s/keys" % tag]]
        assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in d["%s/shapes" % tag]]


def test_split_pack_refuses_weights_out_of_range():
    """the split towers scale the weights' hi half by 2^11 in f16: a folded weight of magnitude >= 31 cannot be packed for
    them; prepare() then keeps the fp32 tower kernels for that network (a warning, not a different result)"""
    import warnings
    from tetris_rl_amd import nets
    w = torch.zeros(16, 16, 3, 3)
    w[3, 5, 1, 1] = 30.0
    nets._pack_conv3x3_split(w)                                   # fits
    w[3, 5, 1, 1] = -40.0
    with pytest.raises(nets.SplitRangeError):
        nets._pack_conv3x3_split(w)
    with pytest.raises(nets.SplitRangeError):
        nets._pack_conv3x3_split_wide(torch.full((32, 32, 3, 3), 64.0), 32)
    with warnings.catch_warnings(record=True) as got:
        warnings.simplefilter("always")
        assert nets._split_or_none(lambda: nets._pack_conv3x3_split(w)) is None
    assert got and "fp32 tower" in str(got[0].message)


def test_keras_weight_port_layout():
    """nets.port_keras_weights (util.py:2369-2464) -- PARITY UNPINNED: Keras is not importable here, so
    nothing Keras computed can be compared.  What can be checked is the layout algebra, on synthetic weights: a
    channels-last ("Keras-side") evaluation of every layer kind with the Keras-layout arrays must equal the PyTorch
    module's evaluation with the ported parameters -- conv kernels (HWIO), dense kernels (in, out), batch-norm statistics,
    and the three heads that read the own-side map flattened in [r, c, k] order instead of PyTorch's [k, r, c]."""
    rng = np.random.default_rng(5)
    mc = nets.AuxBaseResNetConfig(blocks=2, filters=8, opp_hidden=16, own_kernels=3, value_hidden=8, aux_hidden=8)
    f, k = mc.filters, mc.own_kernels
    head_in = k * 400 + 49 + 3 + 1
    r = lambda *s: rng.standard_normal(s).astype(np.float32)
    bn = lambda c: [1 + 0.1 * r(c), 0.1 * r(c), 0.1 * r(c), (0.5 + rng.random(c)).astype(np.float32)]
    kw = {"stem_conv": [r(3, 3, 1, f)], "stem_bn": bn(f)}
    for i in range(mc.blocks):
        kw.update({"block%d_conv1" % i: [r(3, 3, f, f)], "block%d_bn1" % i: bn(f), "block%d_conv2" % i: [r(3, 3, f, f)], "block%d_bn2" % i: bn(f)})
    kw.update({"opp_collapse_conv": [r(1, 1, f, 1)], "opp_collapse_bn": bn(1), "opp_encode_dense": [r(400 + 52, mc.opp_hidden), r(mc.opp_hidden)],
               "opp_encode_bn": bn(mc.opp_hidden), "bias_project": [r(mc.opp_hidden + 53, f), r(f)],
               "own_collapse_conv": [r(1, 1, f, k)], "own_collapse_bn": bn(k), "policy": [0.01 * r(head_in, 11583), r(11583)],
               "value_hidden": [r(head_in, mc.value_hidden), r(mc.value_hidden)], "value_hidden_bn": bn(mc.value_hidden),
               "value": [r(mc.value_hidden, 1), r(1)], "aux_hidden": [r(head_in, mc.aux_hidden), r(mc.aux_hidden)],
               "aux_hidden_bn": bn(mc.aux_hidden), "aux": [r(mc.aux_hidden, 2), r(2)]})
    net = nets.port_keras_weights(kw, mc)
    sd = net.state_dict()
    # conv: channels-last correlation with the HWIO kernel == F.conv2d with the ported OIHW weight
    x = r(2, 40, 10, f)                                                     # (B, H, W, C)
    kern = kw["block1_conv2"][0]
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)))
    want = sum(np.einsum("bhwc,co->bhwo", xp[:, dy:dy + 40, dx:dx + 10, :], kern[dy, dx]) for dy in range(3) for dx in range(3))
    got = torch.nn.functional.conv2d(torch.from_numpy(x.transpose(0, 3, 1, 2).copy()), sd["trunk.1.conv2.weight"], padding=1)
    assert np.abs(got.numpy().transpose(0, 2, 3, 1) - want).max() < 1e-4
    # batch norm: [gamma, beta, mean, var] -> weight, bias, running_mean, running_var
    for name, a in zip(("weight", "bias", "running_mean", "running_var"), kw["block0_bn1"]):
        assert np.array_equal(sd["trunk.0.bn1." + name].numpy(), a)
    # dense: x @ K + b == F.linear(x, W, b)
    v = r(3, mc.opp_hidden + 53)
    assert np.abs(torch.nn.functional.linear(torch.from_numpy(v), sd["bias_project.weight"], sd["bias_project.bias"]).numpy()
                  - (v @ kw["bias_project"][0] + kw["bias_project"][1])).max() < 1e-4
    # heads: Keras flattens the own-side map (B, 40, 10, K) as [r, c, k]; PyTorch flattens (B, K, 40, 10) as [k, r, c]
    own = r(2, 40, 10, k)
    extra = r(2, 53)
    for layer, prefix in (("policy", "policy_head"), ("value_hidden", "value_head.0"), ("aux_hidden", "aux_head.0")):
        keras_in = np.concatenate([own.reshape(2, -1), extra], axis=1)
        torch_in = np.concatenate([own.transpose(0, 3, 1, 2).reshape(2, -1), extra], axis=1)
        want = keras_in @ kw[layer][0] + kw[layer][1]
        got = torch.nn.functional.linear(torch.from_numpy(torch_in), sd[prefix + ".weight"], sd[prefix + ".bias"]).numpy()
        assert np.abs(got - want).max() < 1e-3 * max(1.0, np.abs(want).max()), layer
    # errors name the layer
    bad = dict(kw); del bad["value"]
    with pytest.raises(KeyError, match="value"):
        nets.port_keras_weights(bad, mc)
    bad = dict(kw); bad["stem_conv"] = [r(5, 5, 1, f)]
    with pytest.raises(ValueError, match="stem_conv"):
        nets.port_keras_weights(bad, mc)
    # the ported net is a working inference module of this package
    g, vv = torch.zeros((2, 3, 400)), torch.zeros((3, 105))
    val, pol = net.evaluate_packed(g, vv)
    assert val.shape == (3,) and pol.shape == (3, 11583) and abs(float(pol[0].sum()) - 1) < 1e-4
