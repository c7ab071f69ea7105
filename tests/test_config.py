"""Host-side mirror of the reference's Config (ai.py:62-216): same keywords, defaults, derived
properties and error behaviour.  CPU only."""
import pytest

import tetris_rl_amd as trl


def test_defaults_match_reference_config():
    c = trl.Config()
    # spot values of ai.py:63-138
    assert (c.MAX_ITER, c.CPUCT, c.DPUCT, c.FpuStrategy, c.FpuValue) == (400, 0.75, 1, "reduction", 0.1)
    assert (c.use_root_softmax, c.RootSoftmaxTemp, c.temperature) == (True, 1.1, 0.1)
    assert (c.playout_cap_chance, c.playout_cap_mult, c.DIRICHLET_ALPHA, c.DIRICHLET_S, c.DIRICHLET_EXPLORATION) == (0.25, 5, 0.1, 25, 0.25)
    assert c.ruleset == "s2" and c.move_algorithm == "convolutional" and c.augment_data and not c.save_all
    assert type(c.model_config).__name__ == "AuxBaseResNetConfig"                       # ai.py:83
    assert (c.model_config.blocks, c.model_config.filters, c.model_config.own_kernels) == (8, 32, 4)
    assert (c.value_min, c.value_mid, c.value_max) == (0, 0.5, 1) and c.negate_value(0.25) == 0.75
    t = trl.Config(use_tanh=True)
    assert (t.value_min, t.value_mid, t.value_max) == (-1, 0, 1) and t.negate_value(0.25) == -0.25


def test_unknown_keyword_raises_type_error():
    with pytest.raises(TypeError):
        trl.Config(blocks=10)            # the README's old keyword: not a Config argument


def test_copy_and_dirs(monkeypatch, tmp_path):
    monkeypatch.setenv("TRL_STORAGE", str(tmp_path))
    c = trl.Config(model="pytorch", ruleset="s1", model_version=5.9, data_version=1.3, MAX_ITER=160)
    d = c.copy()
    d.MAX_ITER = 7
    assert c.MAX_ITER == 160 and d.ruleset == "s1"
    assert c.model_dir == "%s/pytorch_models/s1.5.9" % tmp_path
    assert c.data_dir == "%s/data/s1.1.3" % tmp_path
    assert trl.Config(model="keras").model_dir.endswith("/models/s2.7.0")


def test_search_options_reject_what_the_device_lacks():
    with pytest.raises(ValueError, match="Unknown algorithm"):        # the reference's two lossy generators are not built
        trl.search_options(trl.Config(move_algorithm="harddrop"))
    assert trl.search_options(trl.Config(move_algorithm="brute-force")).move_algorithm == 1      # ai.py:84, 414
    assert trl.search_options(trl.Config()).move_algorithm == 0
    with pytest.raises(ValueError):
        trl.search_options(trl.Config(FpuStrategy="nope"))
    o = trl.search_options(trl.Config(MAX_ITER=160, training=True))
    assert o.max_iter == 160 and o.training == 1 and o.use_playout_cap == 1 and abs(o.root_softmax_temp - 1.1) < 1e-12
    assert o.use_tanh == 0 and o.random_start == 0
    o = trl.search_options(trl.Config(use_tanh=True, use_random_starting_moves=True), episodes_base=3, episodes_extra=1)
    assert o.use_tanh == 1 and o.random_start == 1 and (o.episodes_base, o.episodes_extra) == (3, 1)
    c = trl.Config(use_tanh=True)
    assert (c.value_min, c.value_mid, c.value_max, c.negate_value(0.25)) == (-1, 0, 1, -0.25)


def test_seeds_are_fresh_or_derived():
    from tetris_rl_amd import selfplay
    a, b = selfplay.fresh_seed(), selfplay.fresh_seed()
    assert a != b and 0 <= a < 2 ** 63
    d = selfplay.derive_seed(7, "set", 3, 1)
    assert d == selfplay.derive_seed(7, "set", 3, 1) and 0 <= d < 2 ** 63
    assert len({selfplay.derive_seed(7, "set", n, 0) for n in range(50)} | {selfplay.derive_seed(7, "gate", n, 0, 1) for n in range(50)}) == 100
