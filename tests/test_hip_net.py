"""GPU side of rows N1-N3 and D1-D3: features written by the search kernel == game_to_X, the net
on the device within 1e-5 of the reference's logits, and the self-play driver (HIP graph tick)
producing well-formed, deterministic training sets."""
import os

import numpy as np
import pytest
import torch

import tetris_rl_amd as trl
from tetris_rl_amd import records
from test_net_parity import load_net, packed_features, load_formula_net, DEEP, TOL, NET64, load_net64, check_net64, scaled_copy

pytestmark = pytest.mark.gpu

# both tower implementations ship (nets.tower_mode): "split" = k_trunk16s / k_trunk_wide_s (the default), "fp32" =
# k_trunk16 / k_trunk_wide (TRL_TOWER=fp32, and the fallback for weights or activations outside the split kernels' range)
TOWER_MODES = ("split", "fp32")


def _set_mode(net, mode):
    net.tower_mode = mode
    assert net._folded["packed_tower"] is not None and (mode == "fp32" or net._folded["packed_split"] is not None)
    return net


def test_kernel_features_match_game_to_x(golden_dir):
    d = np.load(os.path.join(golden_dir, "env_traces.npz"))
    idx = np.arange(0, d["states"].shape[0], 37)[:128]
    states = d["states"][idx]
    eng = trl.Engine(len(idx), seed=1)
    eng.set_state(states)
    eng.mcts_create(trl.mcts_config(max_iter=4, training=0, auto_reset=0))
    dev = torch.device("cuda", 0)
    grids = torch.zeros((2, len(idx), 400), dtype=torch.float32, device=dev)
    vec = torch.zeros((len(idx), 105), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    eng.mcts_select(grids.data_ptr(), vec.data_ptr())          # first tick: the leaf is the root itself
    leaf, info = eng.mcts_leaf_states()
    g, v = grids.cpu().numpy(), vec.cpu().numpy()
    want_g, want_v = packed_features(leaf)
    assert np.array_equal(g, want_g) and np.array_equal(v, want_v)
    # the root copy restricts previews to 5 (ai.py:306-309); everything else equals the input position
    for i in range(len(idx)):
        a, b = leaf[i].copy(), states[i].copy()
        for pl in (0, 144):
            ql = min(int(b[pl + 139]), 5)
            b[pl + 139] = ql
            q = int.from_bytes(bytes(b[pl + 112:pl + 120]), "little") & ((1 << (4 * ql)) - 1)
            b[pl + 112:pl + 120] = np.frombuffer(q.to_bytes(8, "little"), dtype=np.uint8)
        assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", TOWER_MODES)
@pytest.mark.parametrize("tag,blocks,filters", [("b10f16", 10, 16), ("b2f8", 2, 8)])
def test_net_on_device_matches_reference(golden_dir, tag, blocks, filters, mode):
    d = np.load(os.path.join(golden_dir, "net.npz"))
    net = load_net(d, tag, blocks, filters, device="cuda")
    if filters == 16:
        _set_mode(net, mode)
    elif mode == "fp32":
        pytest.skip("8 filters: no fused tower, one PyTorch path")
    grids, vec = packed_features(d["states"])
    v, logits = net.heads(torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda())
    assert np.abs(torch.sigmoid(v).cpu().numpy() - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.cpu().numpy() - d["%s/logits" % tag]).max() < TOL
    # batch invariance: the same positions inside a batch of 512 give the same numbers to 1e-5
    reps = 64
    vb, lb = net.heads(torch.from_numpy(np.tile(grids, (1, reps, 1))).cuda(), torch.from_numpy(np.tile(vec, (reps, 1))).cuda())
    assert np.abs(lb[:8].cpu().numpy() - d["%s/logits" % tag]).max() < TOL


def _tiny_cfg(**kw):
    return trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=2, filters=8), MAX_ITER=8,
                      training=True, **kw)


def test_make_training_set_is_wellformed_and_deterministic(tmp_path, monkeypatch):
    monkeypatch.setenv("TRL_STORAGE", str(tmp_path / "Storage"))
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=8))
    cfg = _tiny_cfg()
    a = trl.make_training_set(cfg, net, 6, save_stats=False, seed=11)
    b = trl.make_training_set(cfg, net, 6, save_stats=False, seed=11)
    assert len(a) > 0 and len(a) % 4 == 0            # x4 mirror augmentation
    import json
    assert json.dumps(a) == json.dumps(b)
    rec = a[0]
    assert len(rec) == 13 and len(rec[0]) == 40 and len(rec[0][0]) == 10 and len(rec[1]) == 7
    assert rec[11] in (0, 1, 0.5) and len(rec[12]) == 27 and len(rec[12][0]) == 39 and len(rec[12][0][0]) == 11
    s = sum(sum(sum(r) for r in plane) for plane in rec[12])
    assert abs(s - 1.0) < 0.02
    # files: {data_dir}/{n}.txt and logs/stats.jsonl as the reference writes them
    trl.make_training_set(cfg, net, 2, save_game=True, save_stats=True, seed=3)
    assert os.path.isfile(os.path.join(cfg.data_dir, "0.txt"))
    assert os.path.isfile(tmp_path / "Storage" / "logs" / "stats.jsonl")
    data = json.load(open(os.path.join(cfg.data_dir, "0.txt")))
    assert isinstance(data, list) and len(data[0]) == 13
    # the file comes from the text writer (records built as JSON text): same bytes as json.dumps of the list path
    same = trl.make_training_set(cfg, net, 2, save_game=False, save_stats=False, seed=3)
    assert open(os.path.join(cfg.data_dir, "0.txt")).read() == json.dumps(same)
    # write_json=False: the set exists in packed form only (same content as set 0's packed file), and the learner reads it
    from tetris_rl_amd import learner
    trl.make_training_set(cfg, net, 2, save_game=True, save_stats=True, seed=3, write_json=False)
    assert not os.path.exists(os.path.join(cfg.data_dir, "1.txt")) and trl.highest_data_number(cfg) == 1
    p0, p1 = (np.load(os.path.join(learner.packed_dir(cfg), "%d.npz" % k)) for k in (0, 1))
    assert all(np.array_equal(p0[k], p1[k]) for k in p0.files)
    os.remove(os.path.join(cfg.data_dir, "0.txt"))                 # both sets now packed-only
    cfg2 = cfg.copy(); cfg2.batch_size = 16; cfg2.epochs = 1
    entry = learner.load_data_and_train_model(cfg2, trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=8)).cuda())
    assert entry["loss"] > 0


def test_graph_and_eager_ticks_agree():
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=8))
    out = []
    for use_graph in (True, False):
        sp = trl.SelfPlay(_tiny_cfg(), net, 16, seed=5, use_graph=use_graph, auto_reset=False)
        sp.tick(200)
        r = sp.drain()
        r = r[np.lexsort((r["ply"], r["slot"]))]
        out.append(r)
    assert len(out[0]) == len(out[1]) and len(out[0]) > 0
    assert np.array_equal(out[0]["chosen"], out[1]["chosen"]) and np.array_equal(out[0]["visits"], out[1]["visits"])


def test_fused_expand_select_tick_is_bit_identical(monkeypatch):
    """the production tick issues the expand of one simulation and the selection of the next as ONE launch
    (trl_mcts_expand_alphasame_select) inside graphs of eight ticks; the same games come out as with the three-kernel tick,
    one graph launch per tick, and as with eager launches -- also when the states are written under a pending selection"""
    from tetris_rl_amd import selfplay
    torch.manual_seed(0)
    _fused_against_classic(monkeypatch, selfplay, trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=16)).cuda().prepare(),
                           dict(training=True))
    # the value range -1..1 and the random starting moves go through the same launch
    _fused_against_classic(monkeypatch, selfplay, trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=16), use_tanh=True).cuda().prepare(),
                           dict(training=False, use_tanh=True, use_random_starting_moves=True))


def _fused_against_classic(monkeypatch, selfplay, net, options):
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=2, filters=16), MAX_ITER=16, **options)
    out = []
    for fuse, ticks_per_graph, use_graph in ((True, 8, True), (False, 1, True), (False, 1, False)):
        monkeypatch.setattr(selfplay, "FUSE_SELECT", fuse)
        monkeypatch.setattr(selfplay, "GRAPH_TICKS", ticks_per_graph)
        sp = trl.SelfPlay(cfg, net, 48, seed=11, use_graph=use_graph, auto_reset=True)
        assert sp.fused_select == fuse and sp.heads is not None
        sp.engine.set_state(sp.engine.get_state())       # an outside write to the states: the pending selection is redone
        sp.tick(37)                                      # 4 long graphs + 5 single ticks
        sp.tick(300)
        st = sp.stats()
        r = sp.drain()
        r = r[np.lexsort((r["ply"], r["episode"], r["slot"]))]
        out.append((r, st["simulations"]))
        del sp
    for r, sims in out[1:]:
        assert sims == out[0][1] and len(r) == len(out[0][0]) > 48
        for f in ("slot", "episode", "ply", "kind", "winner", "chosen", "saved", "n_children", "state"):
            assert np.array_equal(r[f], out[0][0][f]), f
        valid = np.arange(512)[None, :] < r["n_children"][:, None]
        assert np.array_equal(r["moves"][valid], out[0][0]["moves"][valid]) and np.array_equal(r["visits"][valid], out[0][0]["visits"][valid])


def test_slot_groups_play_the_same_games():
    """SelfPlayGroups: the slots of a job as two independently ticking groups (own engine, stream and graphs each) -- the same
    games, records and counters as one SelfPlay over all slots, through the host drain and through the device drain"""
    from tetris_rl_amd import selfplay, engine
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=2, filters=16)).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=2, filters=16), MAX_ITER=16, training=True)
    streams = np.arange(100, 164, dtype=np.uint32)
    kw = dict(seed=21, streams=streams, auto_reset=True, episodes=(1, 40))       # slots 0..39 play two games, the rest one
    one = trl.SelfPlay(cfg, net, 64, **kw)
    two = trl.SelfPlayGroups(cfg, net, 64, 2, **kw)
    assert [p.n for p in two.parts] == [32, 32] and selfplay.slot_groups(cfg, net, 1024) == 4 and selfplay.slot_groups(cfg, net, 512) == 2 and selfplay.slot_groups(cfg, net, 64) == 1
    got = []
    for sp in (one, two):
        recs = []
        for i in range(6):
            sp.tick(200)
            if i % 2:
                recs.append(sp.drain())
            else:
                t = sp.drain_dev().cpu().numpy()
                recs.append(t.view(engine.RECORD_DTYPE))
        r = np.concatenate(recs)
        got.append((r[np.lexsort((r["kind"], r["ply"], r["episode"], r["slot"]))], sp.stats()))
    (a, sa), (b, sb) = got
    assert len(a) == len(b) > 64 and sa["games"] == sb["games"] > 64 and sa["moves"] == sb["moves"] and sa["simulations"] == sb["simulations"]
    assert (a["episode"] == 1).any() and a["slot"].max() == 63              # second games ran; both groups' slots are there
    for f in ("slot", "episode", "ply", "kind", "winner", "chosen", "saved", "n_children", "state"):
        assert np.array_equal(a[f], b[f]), f
    valid = np.arange(512)[None, :] < a["n_children"][:, None]               # the rest of a row is whatever the buffer held
    assert np.array_equal(a["moves"][valid], b["moves"][valid]) and np.array_equal(a["visits"][valid], b["visits"][valid])


def test_self_play_loop_generates_trains_and_gates(tmp_path, monkeypatch):
    """the whole outer loop of the reference (ai.py:2117-2198) on the device: data set -> learner -> gating
    match -> promotion, twice, with tiny settings"""
    monkeypatch.setenv("TRL_STORAGE", str(tmp_path / "Storage"))
    cfg = _tiny_cfg(training_games=3, battle_games=4, sets_to_train_with=2, batch_size=16)
    cfg.training = False
    n = trl.self_play_loop(cfg, max_sets=2, seed=1)
    assert n == 2 and trl.highest_data_number(cfg) == 1 and trl.highest_model_number(cfg) >= 0
    logs = tmp_path / "Storage" / "logs"
    assert len(open(logs / "training_log.jsonl").read().splitlines()) == 2
    gl = [__import__("json").loads(l) for l in open(logs / "gating_log.jsonl").read().splitlines()]
    assert len(gl) == 2 and all(g["total_games"] == 4 and g["challenger_wins"] + g["best_wins"] <= 4 for g in gl)
    assert trl.highest_model_number(cfg) == sum(1 for g in gl if g["accepted"])


def test_self_play_loop_with_the_default_net_family(tmp_path, monkeypatch):
    """the outer loop with an AuxBaseResNetConfig: the net class follows the model_config as in the reference
    (ai.py:1048-1053), self-play runs on the fused 32-filter tower, the learner adds the auxiliary loss"""
    monkeypatch.setenv("TRL_STORAGE", str(tmp_path / "Storage"))
    mc = trl.AuxBaseResNetConfig(blocks=2, filters=32, opp_hidden=16, own_kernels=4, value_hidden=8, aux_hidden=8)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=8, training=False, training_games=3,
                     battle_games=4, sets_to_train_with=2, batch_size=16)
    assert trl.self_play_loop(cfg, max_sets=1, seed=2) == 1
    best = trl.load_best_model(cfg)
    assert type(best).__name__ == "AuxBaseResNet" and trl.highest_data_number(cfg) == 0
    log = [__import__("json").loads(l) for l in open(tmp_path / "Storage" / "logs" / "training_log.jsonl")]
    assert len(log) == 1 and "aux_loss" in log[0]


@pytest.mark.parametrize("mode", TOWER_MODES)
def test_fused_hip_tower_matches_torch_tower(golden_dir, mode):
    """csrc/trl_net.hip -- k_trunk16s (split f16 operands, the default) and k_trunk16 (fp32 MFMA), activations in LDS --
    against the same tower as PyTorch ops, on random occupancy boards incl. full, empty and checkerboard ones; fp32
    tolerance 2e-5 absolute on outputs of magnitude ~1 (different summation order only)."""
    d = np.load(os.path.join(golden_dir, "net.npz"))
    net = _set_mode(load_net(d, "b10f16", 10, 16, device="cuda"), mode)
    g = torch.Generator().manual_seed(3)
    m = 515
    boards = (torch.rand((m, 1, 40, 10), generator=g) < torch.rand((m, 1, 1, 1), generator=g)).float()
    boards[0] = 0; boards[1] = 1
    boards[2, 0] = torch.from_numpy((np.indices((40, 10)).sum(0) % 2).astype(np.float32))
    boards = boards.cuda()
    net.use_hip_trunk = True
    a = net.trunk(boards)
    net.use_hip_trunk = False
    b = net.trunk(boards)
    net.use_hip_trunk = True
    assert a.shape == b.shape == (m, 400)
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))
    assert float(b.abs().max()) > 0.01


@pytest.mark.parametrize("mode", TOWER_MODES)
@pytest.mark.parametrize("tag,blocks,filters", [("b3f64", 3, 64), ("b2f32", 2, 32)])
def test_wide_fused_tower(golden_dir, tag, blocks, filters, mode):
    """k_trunk_wide_s<32/64> (split) and k_trunk_wide<32/64> (fp32): (1) the reference's logits and values of the fixture
    positions within 1e-5 through the HIP tower, (2) against the PyTorch tower on random boards, (3) the need flags skip boards"""
    d = np.load(os.path.join(golden_dir, "net_wide.npz"))
    net = _set_mode(load_net(d, tag, blocks, filters, device="cuda"), mode)
    assert net.supports_tower_cache
    grids, vec = packed_features(d["states"])
    v, logits = net.heads(torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda())
    assert np.abs(torch.sigmoid(v).cpu().numpy() - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.cpu().numpy() - d["%s/logits" % tag]).max() < TOL
    g = torch.Generator().manual_seed(5)
    m = 301
    boards = (torch.rand((m, 1, 40, 10), generator=g) < torch.rand((m, 1, 1, 1), generator=g)).float()
    boards[0] = 0; boards[1] = 1
    boards = boards.cuda()
    a = net.trunk(boards)
    net.use_hip_trunk = False
    b = net.trunk(boards)
    net.use_hip_trunk = True
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max())) and float(b.abs().max()) > 0.01
    need = (torch.arange(m) % 3 != 0).to(torch.uint8).cuda()
    c = net._trunk_hip(boards, net._folded["packed_tower"], need)
    assert torch.equal(c[need.bool()], a[need.bool()]) and not c[~need.bool()].any()


@pytest.mark.parametrize("mode", TOWER_MODES)
@pytest.mark.parametrize("tag", sorted(DEEP))
def test_deep_and_tanh_nets_on_device(golden_dir, tag, mode):
    """C5's net at its full depth on k_trunk_wide_s<64> / k_trunk_wide<64> (40 convolutions, K = 576) and the tanh
    value head on k_trunk16s / k_trunk16: the reference's values and all 11 583 logits within 1e-5"""
    blocks, filters, salt, use_tanh = DEEP[tag]
    d = np.load(os.path.join(golden_dir, "net_deep.npz"))
    net = _set_mode(load_formula_net(blocks, filters, salt, use_tanh, device="cuda"), mode)
    assert net.use_hip_trunk
    grids, vec = packed_features(np.load(os.path.join(golden_dir, "net.npz"))["states"])
    v, logits = net.heads(torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda())
    v = (torch.tanh(v) if use_tanh else torch.sigmoid(v)).cpu().numpy()
    assert np.abs(v - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.cpu().numpy() - d["%s/logits" % tag]).max() < TOL
    # inside a batch of 256 (other tiles, other workgroups) the same positions give the same numbers
    vb, lb = net.heads(torch.from_numpy(np.tile(grids, (1, 32, 1))).cuda(), torch.from_numpy(np.tile(vec, (32, 1))).cuda())
    assert np.abs(lb[-8:].cpu().numpy() - d["%s/logits" % tag]).max() < TOL


@pytest.mark.parametrize("mode", TOWER_MODES)
@pytest.mark.parametrize("tag", ["base32", "aux32"])
def test_baseresnet_fused_tower(golden_dir, tag, mode):
    """k_trunk_wide_s<32, POST> (split) and trl_resnet_trunk's k_trunk_wide<32, POST> (fp32): the reference's BaseResNet /
    AuxBaseResNet outputs within 1e-5 through the fused tower, and fused == PyTorch tower on a larger random batch"""
    from test_net_parity import load_base32
    d, net = load_base32(golden_dir, tag, device="cuda")
    _set_mode(net, mode)
    grids, vec = packed_features(d["states"])
    tg, tv = torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda()
    v, logits = net.heads(tg, tv)
    assert np.abs(torch.sigmoid(v).cpu().numpy() - d["%s/values" % tag]).max() < TOL
    assert np.abs(logits.cpu().numpy() - d["%s/logits" % tag]).max() < TOL
    if tag == "aux32":
        n = len(d["states"])
        out = net(tg[0].reshape(n, 1, 40, 10), tv[:, :49].reshape(n, 7, 7), tv[:, 49], tv[:, 50], tv[:, 51],
                  tg[1].reshape(n, 1, 40, 10), tv[:, 52:101].reshape(n, 7, 7), tv[:, 101], tv[:, 102], tv[:, 103], tv[:, 104])
        assert len(out) == 3 and np.abs(out[2].cpu().numpy() - d["aux32/aux"]).max() < TOL
    g = torch.Generator().manual_seed(9)
    n = 150
    big = (torch.rand((2, n, 400), generator=g) < torch.rand((2, n, 1), generator=g)).float().cuda()
    bv = tv[torch.arange(n) % tv.shape[0]].contiguous()
    a = net.head_inputs(big, bv)
    net.use_hip_trunk = False
    b = net.head_inputs(big, bv)
    net.use_hip_trunk = True
    assert float((a[1] - b[1]).abs().max()) < 2e-5 * max(1.0, float(b[1].abs().max()))
    assert float((a[0] - b[0]).abs().max()) < 2e-5


def test_legal_only_policy_head_matches_full_softmax_path():
    """trl_mcts_expand_head (logits of the legal moves only, formed in the expand kernel) against the
    reference data flow (softmax over all 11 583 logits, then gather): same root children, priors equal
    to float32 rounding, and the same first move after a whole search."""
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16))
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=32,
                     training=False)
    res = {}
    for mode in ("full", "legal"):
        sp = trl.SelfPlay(cfg, net, 64, seed=9, auto_reset=False, policy_head=mode, use_graph=False)
        sp.tick(1)
        sp.synchronize()
        c, m, v, q, p = sp.engine.mcts_root_children()
        sp.tick(31)
        rec = sp.drain()
        rec = rec[np.argsort(rec["slot"], kind="stable")]
        res[mode] = (c, m, p, rec)
    c0, m0, p0, r0 = res["full"]
    c1, m1, p1, r1 = res["legal"]
    assert np.array_equal(c0, c1)
    for g in range(64):
        n = c0[g]
        assert np.array_equal(m0[g, :n], m1[g, :n])
        assert np.allclose(p0[g, :n], p1[g, :n], rtol=2e-5, atol=1e-9)
    assert len(r0) == len(r1) == 64
    same = int((r0["chosen"] == r1["chosen"]).sum())
    assert same >= 60, "searches diverged in %d of 64 slots" % (64 - same)


def test_baseresnet_selfplay_and_wide_policy_head():
    """the reference's default net family drives the device search; its 853-wide policy-head input
    (small config) goes through the in-kernel legal-logit path and agrees with the full-softmax path"""
    torch.manual_seed(2)
    mc = trl.AuxBaseResNetConfig(blocks=2, filters=8, opp_hidden=16, own_kernels=2, value_hidden=8, aux_hidden=8)
    net = trl.AuxBaseResNet(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=16, training=False)
    res = {}
    for mode in ("full", "legal"):
        sp = trl.SelfPlay(cfg, net, 32, seed=4, auto_reset=False, policy_head=mode)
        sp.tick(1)
        sp.synchronize()
        res[mode] = sp.engine.mcts_root_children()
        sp.tick(15)
        res[mode + "_rec"] = sp.drain()
    c0, m0, _, _, p0 = res["full"]
    c1, m1, _, _, p1 = res["legal"]
    assert np.array_equal(c0, c1)
    for g in range(32):
        assert np.array_equal(m0[g, :c0[g]], m1[g, :c0[g]]) and np.allclose(p0[g, :c0[g]], p1[g, :c0[g]], rtol=2e-5, atol=1e-9)
    assert len(res["full_rec"]) == len(res["legal_rec"]) == 32


def test_c5_configuration_runs():
    """BASELINE config C5 in miniature: 20-block / 64-filter AlphaSame on the fused 64-filter HIP tower,
    MAX_ITER=800 with forced playouts + policy-target pruning; a few ticks of 32 slots"""
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=20, filters=64)
    net = trl.AlphaSame(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=800, training=True,
                     use_forced_playouts_and_policy_target_pruning=True)
    sp = trl.SelfPlay(cfg, net, 32, seed=1)
    assert sp.net._folded["packed_tower"] is not None
    sp.tick(24)
    st = sp.stats()
    assert st["simulations"] == 24 * 32 and st["overflow"] == 0


def test_tower_cache_is_bit_identical():
    """reusing the unchanged board's tower output (trl_mcts_zcache) must not change a single visit count:
    96 slots, 300 ticks (several searches incl. playout-cap short/long ones), with and without the cache"""
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=3, filters=16))
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=3, filters=16), MAX_ITER=24,
                     training=True)
    out = []
    for cache in (True, False):
        sp = trl.SelfPlay(cfg, net, 96, seed=13, tower_cache=cache)
        assert sp.tower_cache == cache
        sp.tick(300)
        r = sp.drain()
        out.append(r[np.lexsort((r["kind"], r["ply"], r["slot"], r["episode"]))])
    a, b = out
    assert len(a) == len(b) > 96
    assert np.array_equal(a["chosen"], b["chosen"]) and np.array_equal(a["visits"], b["visits"])
    assert np.array_equal(a["state"], b["state"])


def test_tower_cache_with_the_default_net_family():
    """BaseResNet on the fused tower: caching the five raw projections per board (trl_mcts_zcache_dim = 2000 floats)
    gives the same games as evaluating both boards at every leaf"""
    torch.manual_seed(0)
    mc = trl.BaseResNetConfig(blocks=2, filters=32, opp_hidden=16, own_kernels=4, value_hidden=8)
    net = trl.BaseResNet(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=24, training=True)
    out = []
    for cache in (True, False):
        sp = trl.SelfPlay(cfg, net, 64, seed=21, tower_cache=cache)
        assert sp.tower_cache == cache and sp.heads is None
        sp.tick(260)
        r = sp.drain()
        out.append(r[np.lexsort((r["kind"], r["ply"], r["slot"], r["episode"]))])
    a, b = out
    assert len(a) == len(b) > 64
    assert np.array_equal(a["chosen"], b["chosen"]) and np.array_equal(a["visits"], b["visits"])
    assert np.array_equal(a["state"], b["state"])


def test_device_heads_match_torch_heads():
    """trl_mcts_expand_alphasame (dense heads + cache assembly inside the expand kernel) against the PyTorch
    heads: same root children, priors and Q to float32 rounding, nearly all first moves equal"""
    torch.manual_seed(0)
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16))
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=32,
                     training=False)
    res = {}
    for dh in (False, True):
        sp = trl.SelfPlay(cfg, net, 64, seed=9, auto_reset=False, device_heads=dh, use_graph=False)
        assert (sp.heads is not None) == dh and sp.tower_cache
        sp.tick(2)
        sp.synchronize()
        c, m, v, q, p = sp.engine.mcts_root_children()
        sp.tick(30)
        rec = sp.drain()
        res[dh] = (c, m, p, q, v, rec[np.argsort(rec["slot"], kind="stable")])
    c0, m0, p0, q0, v0, r0 = res[False]
    c1, m1, p1, q1, v1, r1 = res[True]
    assert np.array_equal(c0, c1)
    for g in range(64):
        n = c0[g]
        assert np.array_equal(m0[g, :n], m1[g, :n]) and np.array_equal(v0[g, :n], v1[g, :n])
        assert np.allclose(p0[g, :n], p1[g, :n], rtol=2e-5, atol=1e-9)
        assert np.allclose(q0[g, :n], q1[g, :n], rtol=0, atol=2e-6)
    assert int((r0["chosen"] == r1["chosen"]).sum()) >= 60


def test_production_tick_matches_reference_priors(golden_dir):
    """The default SelfPlay tick (policy_head="legal", tower cache, heads inside k_expand<2>) next to numbers derived
    from the REFERENCE's outputs on the net.npz positions (not from this build's PyTorch heads): after one tick with
    those positions as roots, every root child's prior equals ai.py:411-443 applied to the reference's logits
    (softmax -> `<= 0 -> 1e-25` -> root softmax exp((ln p - ln p_max) / 1.1) -> normalise, in Python floats), and the
    children's first-play-urgency value equals 1 - value_ref (ai.py:447-453).  Then a second tick: the child the
    reference's PUCT rule picks from those priors is the one the device visited."""
    import math
    import oracle_py as op
    d = np.load(os.path.join(golden_dir, "net.npz"))
    states = d["states"]
    n = len(states)
    net = load_net(d, "b10f16", 10, 16, device="cuda")
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=160,
                     training=False)
    sp = trl.SelfPlay(cfg, net, n, seed=3, auto_reset=False)
    assert sp.policy_head == "legal" and sp.tower_cache and sp.heads is not None and sp.native_graph   # the production path
    sp.engine.set_state(states)
    sp.tick(1)
    sp.synchronize()
    counts, moves, visits, q, p = sp.engine.mcts_root_children()
    logits = d["b10f16/logits"].astype(np.float32)
    want_children = []
    for g in range(n):
        legal = op.OracleGame(state=states[g]).legal_moves()
        assert counts[g] == len(legal) > 0 and np.array_equal(moves[g, :counts[g]], legal)
        probs = torch.softmax(torch.from_numpy(logits[g]), dim=0).numpy()              # evaluate_pytorch, ai.py:1315-1318
        probs[probs <= 0] = 1e-25
        idx = [((int(m) >> 10) & 31) * 429 + ((int(m) >> 4) & 63) * 11 + (int(m) & 15) for m in legal]
        pol = [probs[i] for i in idx]                                                   # np.float32 scalars, as in the reference
        log_max, inv_temp = math.log(max(pol)), 1.0 / cfg.RootSoftmaxTemp
        pol = [math.exp((math.log(x) - log_max) * inv_temp) for x in pol]
        s = sum(pol)
        want = np.array([x / s for x in pol])
        want_children.append(want)
        got = p[g, :counts[g]]
        assert np.abs(got - want).max() < 1e-6 and np.abs(got / want - 1).max() < 2e-5, (g, np.abs(got / want - 1).max())
        v_ref = float(d["b10f16/values"][g])
        assert np.abs(q[g, :counts[g]] - (1.0 - v_ref)).max() < TOL                   # max(value_min, 1 - v), ai.py:447-453
        assert (visits[g, :counts[g]] == 0).all()
    # second simulation: selection by Q + c * P * sqrt(N_parent) / d over unvisited children, last maximum wins
    # (ai.py:346-393); skipped where the reference's own top two scores are closer than the prior tolerance
    sp.tick(1)
    sp.synchronize()
    _, _, visits2, _, _ = sp.engine.mcts_root_children()
    checked = 0
    for g in range(n):
        score = (1.0 - float(d["b10f16/values"][g])) + cfg.CPUCT * want_children[g] * math.sqrt(1.0) / cfg.DPUCT
        order = np.sort(score)
        if order[-1] - order[-2] < 1e-6:
            continue
        best = int(np.nonzero(score == score.max())[0][-1])
        assert visits2[g, :counts[g]].sum() == 1 and visits2[g, best] == 1
        checked += 1
    assert checked >= 6


# ---- round 3: every shipped tower path at the size it ships at, on positions chosen for the activations' magnitude ----
@pytest.mark.parametrize("mode", TOWER_MODES)
@pytest.mark.parametrize("tag", NET64)
def test_64_positions_on_device(golden_dir, tag, mode):
    """tests/golden/net64.npz: the reference's value and logits on 64 positions -- tall stacks, garbage-heavy, nearly full
    and nearly empty boards (the split-f16 scheme's error depends on the activations' magnitude) -- for C3's net, its tanh
    variant, C5's 20 x 64 net and a 32-filter net, through both tower implementations: within 1e-5"""
    net, use_tanh = load_net64(golden_dir, tag, device="cuda")
    _set_mode(net, mode)
    err = check_net64(golden_dir, tag, net, use_tanh, device="cuda")
    assert err < TOL
    assert trl.engine.tower_overflow(0) == 0


def test_weights_outside_the_split_range_take_the_fp32_towers(golden_dir):
    """a net whose folded conv weights reach |w| >= 31 (the split kernels form 2^11 hi_w in f16) must run on the fp32
    kernels -- said with a RuntimeWarning when the weights are packed -- and meet the same 1e-5: b10f16 with one block
    rescaled by 2^10 (the same function; powers of two are exact, so the reference's own outputs are net.npz's)"""
    d = np.load(os.path.join(golden_dir, "net.npz"))
    base = load_net(d, "b10f16", 10, 16, device="cpu")
    big = scaled_copy(base, "weight").cuda()
    with pytest.warns(RuntimeWarning, match="fp32 tower kernels"):
        big.prepare()
    fd = big._folded
    assert fd["packed_tower"] is not None and fd["packed_split"] is None and big.tower_mode == "split"
    assert float(fd["blocks"][3][2].abs().max()) >= 31.0
    grids, vec = packed_features(d["states"])
    v, logits = big.heads(torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda())
    assert np.abs(torch.sigmoid(v).cpu().numpy() - d["b10f16/values"]).max() < TOL
    assert np.abs(logits.cpu().numpy() - d["b10f16/logits"]).max() < TOL
    check_net64(golden_dir, "b10f16", big, False, device="cuda")
    # inside the search: the production tick of such a net is the fp32 kernel (k_trunk16) and plays on
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=16, training=True)
    sp = trl.SelfPlay(cfg, big, 32, seed=5)
    sp.tick(64)
    st = sp.stats()
    assert st["simulations"] == 64 * 32 and st["overflow"] == 0 and st["tower_overflow"] == 0


def test_activations_outside_the_split_range_are_reported_and_recomputed(golden_dir):
    """weights < 31 but activations > 65 504 (no finite f16 hi half): the split kernel counts the board
    (trl_tower_overflow), trunk() warns, switches the net to the fp32 kernels and recomputes -- 1e-5 against the
    reference either way; inside a captured tick the count shows up in SelfPlay.stats() and make_training_set refuses the
    set.  b10f16 with block 3's first batch norm x 2^17 and its first conv / 2^17: the same function, bit for bit."""
    d = np.load(os.path.join(golden_dir, "net.npz"))
    base = load_net(d, "b10f16", 10, 16, device="cpu")
    hot = scaled_copy(base, "activation").cuda().prepare()
    assert hot._folded["packed_split"] is not None and hot.tower_mode == "split"       # the weights fit
    grids, vec = packed_features(d["states"])
    tg, tv = torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda()
    # the kernel alone: finite garbage or NaN, and the counter says so
    trl.engine.tower_overflow(0)
    hot._trunk_hip(tg.reshape(-1, 1, 40, 10), hot._folded["packed_tower"])
    torch.cuda.synchronize()
    assert 8 <= trl.engine.tower_overflow(0) <= 4 * 2 * len(d["states"])              # counted per wavefront: up to 4 per board
    assert trl.engine.tower_overflow(0) == 0                                           # reading resets
    with pytest.warns(trl.nets.SplitActivationRange):
        v, logits = hot.heads(tg, tv)
    assert hot.tower_mode == "fp32"
    assert np.abs(torch.sigmoid(v).cpu().numpy() - d["b10f16/values"]).max() < TOL
    assert np.abs(logits.cpu().numpy() - d["b10f16/logits"]).max() < TOL
    # the captured tick cannot stop to look: the counter is read at the next synchronisation point
    hot.tower_mode = "split"
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=16, training=True)
    sp = trl.SelfPlay(cfg, hot, 16, seed=5)
    sp.tick(8)
    assert sp.stats()["tower_overflow"] > 0
    with pytest.raises(trl.EngineError, match="tower_overflow"):
        trl.make_training_set(cfg, hot, 4, save_stats=False, seed=1)
    hot.tower_mode = "fp32"
    assert len(trl.make_training_set(cfg, hot, 4, save_stats=False, seed=1)) > 0


def _sorted_records(sp):
    r = sp.drain()
    return r[np.lexsort((r["kind"], r["ply"], r["episode"], r["slot"]))]


def test_production_tick_at_c3_size(monkeypatch):
    """BASELINE config C3 as the bench times it -- 1024 slots, AlphaSame(10, 16) on k_trunk16s, training=True, the fused
    expand + select launch inside eight-tick graphs, the leaves' legal-move search in the tower launch, four slot groups on
    four streams -- for 480 ticks, against the same job run as two groups and as one group, with the legal-move search in
    the selection (TRL_LEGAL_IN_TOWER=0), with the three-kernel tick (TRL_FUSE_SELECT=0) and with one tick per graph (TRL_GRAPH_TICKS=1):
    identical records, no overflow of any kind, and the invariants of a search (ai.py:299-659) on every record."""
    from tetris_rl_amd import selfplay
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=10, filters=16)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
    n, ticks, seed = 1024, 480, 20260630
    out = []
    # (groups, fused expand + select, ticks per graph, the leaves' legal-move search inside the tower launch)
    for groups, fuse, per_graph, in_tower in ((4, True, 8, True), (2, True, 8, True), (1, True, 8, True), (1, True, 8, False),
                                              (1, False, 8, False), (1, True, 1, True)):
        monkeypatch.setattr(selfplay, "FUSE_SELECT", fuse)
        monkeypatch.setattr(selfplay, "GRAPH_TICKS", per_graph)
        monkeypatch.setattr(selfplay, "LEGAL_IN_TOWER", in_tower)
        kw = dict(seed=seed, auto_reset=True, record_capacity=1 << 15)
        sp = trl.SelfPlayGroups(cfg, net, n, groups, **kw) if groups > 1 else trl.SelfPlay(cfg, net, n, **kw)
        part = getattr(sp, "parts", [sp])[0]
        assert part.fused_select == fuse and part.native_graph and part.heads is not None and part.net.tower_mode == "split"
        assert part.legal_in_tower == (fuse and in_tower)
        sp.tick(ticks)
        st = sp.stats()
        assert st["overflow"] == 0 and st["records_dropped"] == 0 and st["tower_overflow"] == 0
        assert st["simulations"] == n * ticks
        out.append((_sorted_records(sp), st))
        del sp, part
        torch.cuda.empty_cache()
    a, sa = out[0]
    moves = a[a["kind"] == trl.REC_MOVE]
    assert len(moves) >= 2 * n                                   # every slot finished searches, short and long ones
    nch = moves["n_children"].astype(np.int64)
    valid = np.arange(512)[None, :] < nch[:, None]
    vis = (moves["visits"].astype(np.int64) * valid).sum(axis=1)
    assert set(np.unique(vis)) <= {399, 79}                      # root children's visits = simulations - 1 (playout cap 400 / 80)
    assert (moves["saved"] == (vis == 399)).all()                # only the full searches are saved (ai.py:1613)
    assert ((moves["moves"] == moves["chosen"][:, None]) & valid).any(axis=1).all()      # the move played is a root child
    for b, sb in out[1:]:
        assert sb["moves"] == sa["moves"] and sb["games"] == sa["games"] and len(b) == len(a)
        for f in ("slot", "episode", "ply", "kind", "winner", "chosen", "saved", "n_children", "state"):
            assert np.array_equal(a[f], b[f]), f
        va = np.arange(512)[None, :] < a["n_children"][:, None]
        assert np.array_equal(a["moves"][va], b["moves"][va]) and np.array_equal(a["visits"][va], b["visits"][va])


def test_whole_searches_with_the_real_net_match_the_reference(golden_dir):
    """32 reference searches -- MCTS(config, game, net) at MAX_ITER = 160 with the real 10 x 16 net through
    evaluate_pytorch (ai.py:299-659, 1296-1319), on tall / garbage-heavy / sparse positions (net64.npz) -- against the
    production tick (split-f16 tower, tower cache, heads inside the search kernel, fused expand + select).  Logits agree
    to ~1e-6, but PUCT is an arg-max: where two children's scores are closer than that, one simulation goes to the other
    child and the subtrees differ from there on.  Stated bar: the same set of root children; the same move wherever the
    reference's most visited child leads the runner-up by more than 2 visits; and an L1 distance between the visit
    vectors of at most 16 of the 159 visits in every search, at most 4 on average."""
    d = np.load(os.path.join(golden_dir, "net64.npz"))
    pos = d["search_pos"]
    states = d["states"][pos]
    n = len(pos)
    net = load_net(np.load(os.path.join(golden_dir, "net.npz")), "b10f16", 10, 16, device="cuda")
    cfg = trl.Config(visual=False, model="pytorch", model_config=trl.AlphaSameConfig(blocks=10, filters=16), MAX_ITER=160,
                     training=False)
    sp = trl.SelfPlay(cfg, net, n, seed=int(np.load(os.path.join(golden_dir, "env_traces.npz"))["seed"]), auto_reset=False)
    assert sp.fused_select and sp.native_graph and sp.net.tower_mode == "split"
    sp.engine.set_state(states)
    sp.tick(160)
    st = sp.stats()
    assert st["tower_overflow"] == 0 and st["overflow"] == 0
    r = sp.drain()
    r = r[(r["kind"] == trl.REC_MOVE) & (r["ply"] == np.array([int.from_bytes(bytes(s[300:304]), "little") for s in states])[r["slot"]])]
    r = r[np.argsort(r["slot"], kind="stable")]
    assert len(r) == n                                           # every slot's first search is the fixture's
    off = np.concatenate([[0], np.cumsum(d["search_child_counts"])])
    l1, decided, same = [], 0, 0
    for g in range(n):
        k = int(d["search_child_counts"][g])
        ref_moves, ref_vis = d["search_child_moves"][off[g]:off[g + 1]], d["search_child_visits"][off[g]:off[g + 1]]
        assert int(r["n_children"][g]) == k and np.array_equal(r["moves"][g][:k], ref_moves)
        mine = r["visits"][g][:k].astype(np.int64)
        assert mine.sum() == ref_vis.sum() == 159
        l1.append(int(np.abs(mine - ref_vis).sum()))
        top = np.sort(ref_vis)[::-1]
        if k == 1 or top[0] - top[1] > 2:
            decided += 1
            assert int(r["chosen"][g]) == int(d["search_move"][g]), (g, l1[-1])
        same += int(r["chosen"][g]) == int(d["search_move"][g])
    print("real-net searches: L1 visit distance max %d mean %.2f; %d of %d decided searches, %d same moves" %
          (max(l1), float(np.mean(l1)), decided, n, same))
    assert decided >= 8
    assert max(l1) <= 16 and float(np.mean(l1)) <= 4.0


def test_slot_group_streams_run_side_by_side():
    """selfplay.concurrent_streams: the streams the slot groups tick on were checked to overlap pairwise (two HIP streams
    can share a hardware queue, and then the groups' kernels take turns); drain_dev(then_tick=...) hands the records over
    and lets the group go on ticking -- the same records as drain followed by tick"""
    import time
    from tetris_rl_amd import selfplay, engine
    dev = torch.device("cuda", 0)
    streams = selfplay.concurrent_streams(torch, dev, 4)
    assert len(streams) == 4 and len({s.cuda_stream for s in streams}) == 4
    cycles = int(2.0e6)                                          # ~1 ms of spinning per stream
    torch.cuda._sleep(1000); torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        torch.cuda._sleep(cycles)
    streams[0].synchronize()
    one = time.perf_counter() - t0
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    for s in streams:
        s.synchronize()
    four = time.perf_counter() - t0
    assert four < 2.0 * one, (one, four)                         # four spins side by side, not one after the other
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=16, training=True)
    out = []
    for overlap in (False, True):
        sp = trl.SelfPlayGroups(cfg, net, 64, 2, seed=21)
        recs = []
        sp.tick(96)
        for i in range(3):
            if overlap:
                t = sp.drain_dev(then_tick=96 if i < 2 else 0)
            else:
                t = sp.drain_dev()
                if i < 2:
                    sp.tick(96)
            recs.append(t.cpu().numpy().view(engine.RECORD_DTYPE))
        r = np.concatenate(recs)
        out.append((r[np.lexsort((r["kind"], r["ply"], r["episode"], r["slot"]))], sp.stats()))
    (a, sa), (b, sb) = out
    assert sa["simulations"] == sb["simulations"] == 64 * 96 * 3 and len(a) == len(b) > 64
    for f in ("slot", "episode", "ply", "kind", "winner", "chosen", "saved", "n_children", "state"):
        assert np.array_equal(a[f], b[f]), f


def test_records_leave_without_a_host_round_trip():
    """SelfPlay.drain_async / take (trl_mcts_drain_records_async): the next chunk of ticks is queued behind the drain, the
    records of the previous chunk are picked up on a side stream -- the same records, slot numbers and counters as the
    drain that waits for the stream; and what does not fit the caller's buffer is counted as dropped"""
    from tetris_rl_amd import engine, selfplay
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=16, training=True)
    size = engine.RECORD_DTYPE.itemsize

    def sorted_records(parts):
        r = np.concatenate(parts)
        return r[np.lexsort((r["kind"], r["ply"], r["episode"], r["slot"]))]

    sp = trl.SelfPlayGroups(cfg, net, 64, 2, seed=21)
    plain = []
    for i in range(4):
        sp.tick(96)
        plain.append(sp.drain())
    a, sa = sorted_records(plain), sp.stats()

    sp = trl.SelfPlayGroups(cfg, net, 64, 2, seed=21)
    piped, counters, pending = [], [], None
    for i in range(4):
        sp.tick(96)
        ticket = sp.drain_async()
        if pending is not None:
            t, st = sp.take(pending)
            piped.append(t.cpu().numpy().view(engine.RECORD_DTYPE))
            counters.append(st)
        pending = ticket
    with pytest.raises(RuntimeError):
        sp.parts[0].drain_async(), sp.parts[0].drain_async()           # a third ticket with two outstanding
    sp.parts[0]._acc_busy = [False, False]                              # (the refused draw changed nothing else)
    t, st = sp.take(pending)
    piped.append(t.cpu().numpy().view(engine.RECORD_DTYPE))
    counters.append(st)
    b, sb = sorted_records(piped), sp.stats()
    assert len(a) == len(b) > 64 and a.tobytes() == b.tobytes()
    assert [c["ticks"] for c in counters] == [96, 192, 288, 384]
    assert counters[-1]["simulations"] == sa["simulations"] == sb["simulations"] == 64 * 96 * 4
    assert counters[-1]["moves"] == sa["moves"] and counters[-1]["games"] == sa["games"]
    assert sorted(set(b["slot"].tolist())) == list(range(64))          # the second group's records carry global slots

    # a caller's buffer with room for 5 records: 5 arrive, the rest are counted as dropped; appends go behind the head
    one = trl.SelfPlay(cfg, net, 32, seed=21)
    one.tick(96)
    buf = torch.zeros(5 * size, dtype=torch.uint8, device="cuda")
    head = torch.zeros(selfplay.DRAIN_HEAD_WORDS, dtype=torch.int64, device="cuda")
    one.engine.mcts_drain_records_async(buf.data_ptr(), 5, head.data_ptr(), 1000)
    one.synchronize()
    h = head.cpu().numpy()
    ref = trl.SelfPlay(cfg, net, 32, seed=21)
    ref.tick(96)
    want = ref.drain()
    assert len(want) > 5 and h[0] == 5 and h[1 + 3] == len(want) - 5
    got = buf.cpu().numpy().view(engine.RECORD_DTYPE)
    assert set((got["slot"].astype(np.int64) - 1000).tolist()) <= set(range(32))      # slot_offset was added
    assert one.drain().size == 0                                       # the pending count was cleared


def test_legal_moves_in_the_tower_launch_keep_their_order():
    """trl_mcts_legal_in_tower: a selection that left its legal-move search to the tower launch must be followed by
    trl_mcts_tower16_legal -- an expand or another selection before it is refused (TRL_E_STATE), and so is the tower call
    without the mode; after the tower launch the leaves' move lists are the ones the selection itself would have made"""
    from tetris_rl_amd import engine, selfplay
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=16, training=True)
    a = trl.SelfPlay(cfg, net, 48, seed=5, use_graph=False)
    b = trl.SelfPlay(cfg, net, 48, seed=5, use_graph=False)
    assert net.supports_tower_legal
    with pytest.raises(engine.EngineError):                        # the mode is off
        net.tower_legal(a.engine, a.grids, a.need, out=a.z)
    a.engine.mcts_legal_in_tower(True)
    for sp in (a, b):
        with torch.cuda.stream(sp.stream):
            sp.engine.mcts_select_need(sp.grids.data_ptr(), sp.vec.data_ptr(), sp.need.data_ptr())
    with torch.cuda.stream(a.stream):
        with pytest.raises(engine.EngineError):                    # the leaves have no move lists yet
            a.engine.mcts_expand_alphasame(a.z.data_ptr(), a.vec.data_ptr(), a.heads)
        with pytest.raises(engine.EngineError):
            a.engine.mcts_select_need(a.grids.data_ptr(), a.vec.data_ptr(), a.need.data_ptr())
        with pytest.raises(engine.EngineError):
            a.engine.mcts_legal_in_tower(False)
    for t in range(40):
        with torch.cuda.stream(a.stream):
            net.tower_legal(a.engine, a.grids, a.need, out=a.z)
            a.engine.mcts_expand_alphasame_select(a.z.data_ptr(), a.vec.data_ptr(), a.heads, a.grids.data_ptr(), a.need.data_ptr())
        with torch.cuda.stream(b.stream):
            net.tower_need(b.grids, b.need, out=b.z)
            b.engine.mcts_expand_alphasame_select(b.z.data_ptr(), b.vec.data_ptr(), b.heads, b.grids.data_ptr(), b.need.data_ptr())
    with torch.cuda.stream(a.stream):
        net.tower_legal(a.engine, a.grids, a.need, out=a.z)        # the pending selection's lists
    a.synchronize(), b.synchronize()
    sa, ia = a.engine.mcts_leaf_states()
    sb, ib = b.engine.mcts_leaf_states()
    assert np.array_equal(sa, sb) and np.array_equal(ia, ib)
    ra, rb = a.drain(), b.drain()                                  # (in the order the slots' workgroups got there)
    ra, rb = (r[np.lexsort((r["kind"], r["ply"], r["episode"], r["slot"]))] for r in (ra, rb))
    assert len(ra) == len(rb) > 48 and ra.tobytes() == rb.tobytes()


def test_boards_the_tower_has_seen_are_not_evaluated_again(monkeypatch):
    """The search's board table (DevMcts::btab): a leaf whose mover's board was evaluated earlier in the same search --
    the same placement under another reply of the opponent -- takes that tower output from the per-state cache instead of
    the tower.  Same function of the same board, so the games are the same byte for byte as with TRL_BOARD_TABLE=0, and
    about half of a 160-simulation search's boards are such repeats; with the default net family (k_zcache flow) as well."""
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=10, filters=16)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
    out = []
    for table in ("1", "0"):
        monkeypatch.setenv("TRL_BOARD_TABLE", table)
        sp = trl.SelfPlay(cfg, net, 192, seed=20260630, auto_reset=True, record_capacity=1 << 14)
        assert sp.fused_select and sp.legal_in_tower
        sp.tick(520)
        st = sp.stats()
        assert st["overflow"] == 0 and st["records_dropped"] == 0 and st["tower_overflow"] == 0
        out.append((_sorted_records(sp), st))
    (a, sa), (b, sb) = out
    assert sb["boards_reused"] == 0 and sa["boards_reused"] > 0.3 * sa["simulations"], (sa, sb)
    assert sa["moves"] == sb["moves"] > 192 and a.tobytes() == b.tobytes()

    mc = trl.BaseResNetConfig(blocks=2, filters=32, opp_hidden=16, own_kernels=4, value_hidden=8)
    net = trl.BaseResNet(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=48, training=True)
    out = []
    for table in ("1", "0"):
        monkeypatch.setenv("TRL_BOARD_TABLE", table)
        sp = trl.SelfPlay(cfg, net, 64, seed=21)
        assert sp.tower_cache and sp.heads is None
        sp.tick(300)
        st = sp.stats()
        out.append((_sorted_records(sp), st))
    (a, sa), (b, sb) = out
    assert sb["boards_reused"] == 0 and sa["boards_reused"] > 0
    assert len(a) == len(b) > 64 and a.tobytes() == b.tobytes()


def test_board_table_in_long_searches_with_the_wide_tower(monkeypatch):
    """the board table at C5's search length (MAX_ITER = 800: searches of 400 and 2000 simulations, forced playouts and
    policy-target pruning) with a 64-filter net -- the tower launch that takes the live boards first -- against the same
    job with every board sent to the tower: the same records, and well over half of the boards served from the table"""
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=64)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=800, training=True,
                     use_forced_playouts_and_policy_target_pruning=True)
    out = []
    for table in ("1", "0"):
        monkeypatch.setenv("TRL_BOARD_TABLE", table)
        sp = trl.SelfPlay(cfg, net, 48, seed=11, auto_reset=True)
        assert sp.fused_select and not sp.legal_in_tower
        sp.tick(2600)
        st = sp.stats()
        assert st["overflow"] == 0 and st["records_dropped"] == 0 and st["tower_overflow"] == 0
        out.append((_sorted_records(sp), st))
    (a, sa), (b, sb) = out
    assert sb["boards_reused"] == 0 and sa["boards_reused"] > 0.5 * sa["simulations"], (sa, sb)
    assert sa["moves"] == sb["moves"] >= 48 and a.tobytes() == b.tobytes()


def test_legal_job_of_the_tower_launch_on_synthetic_boards():
    """the legal-move job inside the tower launch (closure on two waves, the speculative ordered walk of a T slot on the other
    two: block_movegen_spec) on 4 096 synthetic boards, half of them with a T to place: the root's children after one tick
    are the oracle's legal moves of the position, in its order"""
    import oracle_py as op
    from test_hip_parity import _state_for_case
    rng = np.random.default_rng(77)
    n = 4096
    states = []
    for i in range(n):
        rows = np.zeros(40, dtype=np.uint16)
        h = int(rng.integers(3, 21))
        p_fill = rng.uniform(0.35, 0.85)
        well = int(rng.integers(0, 10)) if rng.random() < 0.5 else -1
        for r in range(40 - h, 40):
            cells = rng.random(10) < (p_fill * (0.6 + 0.4 * (r - (40 - h)) / max(h - 1, 1)))
            if well >= 0 and rng.random() < 0.85:
                cells[well] = False
            if cells.all():
                cells[int(rng.integers(0, 10))] = False
            rows[r] = int(sum(1 << c for c in range(10) if cells[c]))
        piece = 6 if rng.random() < 0.5 else int(rng.integers(0, 7))
        st = _state_for_case(rows, piece, int(rng.choice([7, 6, int(rng.integers(0, 7))])), int(rng.integers(0, 7)))
        st[144:144 + 80] = st[0:80]                                    # the other player: any valid board
        st[144 + 137] = int(rng.integers(0, 7))
        states.append(st)
    states = np.stack(states)
    torch.manual_seed(0)
    mc = trl.AlphaSameConfig(blocks=2, filters=16)
    net = trl.AlphaSame(mc).cuda().prepare()
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=16, training=False)
    sp = trl.SelfPlay(cfg, net, n, seed=5, auto_reset=False, max_nodes=4096)
    assert sp.fused_select and sp.legal_in_tower
    sp.engine.set_state(states)
    sp.tick(1)
    sp.synchronize()
    counts, moves, visits, q, p = sp.engine.mcts_root_children()
    live = 0
    for g in range(n):
        want = op.OracleGame(state=states[g]).legal_moves()
        assert counts[g] == len(want) and np.array_equal(moves[g, :counts[g]], want), "synthetic board %d" % g
        live += int(len(want) > 0)
    assert live > 0.8 * n and sp.stats()["overflow"] == 0
