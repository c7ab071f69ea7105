"""Learner on ROCm: train_network_pytorch / load_data_and_train_model of the
reference (ai.py:1139-1220, 1871-1921) for the PyTorch net families, on the training sets the rollout
engine writes.  Plain PyTorch (autograd, AdamW); the nets' training-mode forwards apply the layers in
the reference's order so that loss values and updated weights match the reference step for step.
"""
import json
import os
import random

import numpy as np
import torch
from torch import nn
import torch.nn.functional as F

from . import nets, selfplay

ROWS, COLS, POLICY_SIZE = 40, 10, 27 * 39 * 11


def compute_aux_targets(own_grid_batch):
    """architectures.py:285-305: [holes, aggregate height] / (ROWS * COLS) of the own board"""
    g = own_grid_batch.squeeze(1)
    filled = (g > 0.5).float()
    above = (torch.cumsum(filled, dim=1) > 0).float()
    holes = ((1.0 - filled) * above).sum(dim=(1, 2))
    any_filled = filled.sum(dim=1) > 0
    top = filled.argmax(dim=1)
    heights = torch.where(any_filled, (ROWS - top).float(), torch.zeros_like(top, dtype=torch.float))
    return torch.stack([holes / float(ROWS * COLS), heights.sum(dim=1) / float(ROWS * COLS)], dim=1)


# ---- training-mode forwards (layer order of architectures.py:119-142, 230-271, 325-353) -------------------
def _alphasame_train(net, a_grid, a_pieces, a_b2b, a_combo, a_garbage, o_grid, o_pieces, o_b2b, o_combo, o_garbage, color):
    p = net.model_config.dropout

    def grid(x):
        x = net.conv1(x)
        for b in net.res_blocks:
            h = b.conv_block1["2"](torch.relu(b.conv_block1["0"](x)))
            h = b.conv_block2["3"](torch.relu(F.dropout(b.conv_block2["0"](h), p, net.training)))
            x = h + x
        x = torch.relu(net.batchnorm1(x))
        return torch.relu(net.batchnorm2(net.kernel1(x))).flatten(1)

    a = grid(a_grid)
    o = torch.relu(net.osidedense["1"](net.osidedense["0"](grid(o_grid))))
    x = torch.cat([a, a_pieces.flatten(1), a_b2b.unsqueeze(1), a_combo.unsqueeze(1), a_garbage.unsqueeze(1),
                   o, o_pieces.flatten(1), o_b2b.unsqueeze(1), o_combo.unsqueeze(1), o_garbage.unsqueeze(1), color.unsqueeze(1)], dim=1)
    v = net.value_head["3"](torch.relu(net.value_head["1"](net.value_head["0"](x))))
    v = F.dropout(v, p, net.training)
    return (torch.tanh(v) if net.use_tanh else torch.sigmoid(v)), net.policy_head(x)


def _baseresnet_train(net, a_grid, a_pieces, a_b2b, a_combo, a_garbage, o_grid, o_pieces, o_b2b, o_combo, o_garbage, color):
    def grid(x):
        x = torch.relu(net.stem["1"](net.stem["0"](x)))
        for b in net.trunk:
            out = torch.relu(b.bn1(b.conv1(x)))
            x = torch.relu(b.bn2(b.conv2(out)) + x)
        return x

    a_feat, o_feat = grid(a_grid), grid(o_grid)
    o_col = torch.relu(net.opp_collapse["1"](net.opp_collapse["0"](o_feat))).flatten(1)
    opp = torch.relu(net.opp_encode["1"](net.opp_encode["0"](torch.cat(
        [o_col, o_pieces.flatten(1), o_b2b.unsqueeze(1), o_combo.unsqueeze(1), o_garbage.unsqueeze(1)], dim=1))))
    own = [a_pieces.flatten(1), a_b2b.unsqueeze(1), a_combo.unsqueeze(1), a_garbage.unsqueeze(1), color.unsqueeze(1)]
    bias = net.bias_project(torch.cat([opp] + own, dim=1))
    flat = torch.relu(net.own_collapse["1"](net.own_collapse["0"](a_feat + bias.view(-1, bias.size(1), 1, 1)))).flatten(1)
    hx = torch.cat([flat] + own, dim=1)
    v = net.value_head["3"](torch.relu(net.value_head["1"](net.value_head["0"](hx))))
    v = torch.tanh(v) if net.use_tanh else torch.sigmoid(v)
    out = (v, net.policy_head(hx))
    if net.has_aux:
        out += (torch.sigmoid(net.aux_head["3"](torch.relu(net.aux_head["1"](net.aux_head["0"](hx))))),)
    return out


def forward_train(net, *features):
    """autograd-enabled forward with the reference's layer order (BN batch statistics / dropout follow net.training)"""
    if isinstance(net, nets.AlphaSame):
        return _alphasame_train(net, *features)
    if isinstance(net, nets.BaseResNet):
        return _baseresnet_train(net, *features)
    raise TypeError("no training forward for %r" % type(net).__name__)


# ---- the training step (ai.py:1139-1220) -------------------------------------------------------------------------
# Same arithmetic in the same order as the reference -- the tests compare losses and every updated weight with the
# reference's own step -- organised as: columns -> batches (the reference's DataLoader, because its shuffling draws from
# torch's global generator and the order of the batches is part of the result) -> objective terms -> running means.
class _Columns:
    """the 13 feature columns of a training set as tensors (board columns as float [n, 1, 40, 10])"""

    def __init__(self, source):
        if isinstance(source, tuple):                          # records.PackedSet.features(): arrays, no detour through lists
            cols = [torch.from_numpy(np.ascontiguousarray(c)) for c in source]
        else:                                                  # the reference's list of 13-field records
            cols = [torch.tensor(list(c)) for c in zip(*source)]
        self.tensors = [c.unsqueeze(1).to(torch.float) if (c.dim() == 3 and tuple(c.shape[1:]) == (ROWS, COLS)) else c for c in cols]

    def batches(self, config):
        return torch.utils.data.DataLoader(torch.utils.data.TensorDataset(*self.tensors), batch_size=config.batch_size,
                                           shuffle=config.shuffle)


class _Means:
    """per-term sums over the optimiser steps of one call, reported as means"""

    def __init__(self, names):
        self.sums, self.steps = {n: 0.0 for n in names}, 0

    def add(self, **terms):
        for n, v in terms.items():
            self.sums[n] += v.item()
        self.steps += 1

    def mean(self, name):
        return self.sums[name] / max(self.steps, 1)


def _objective(model, inputs, z, pi, aux_weight):
    """(total, value term, policy term, aux term or None): MSE on the value, cross-entropy against the visit distribution,
    MSE on the auxiliary head (holes / height of the own board, weighted) for the Aux nets"""
    out = forward_train(model, *inputs)
    value_term = F.mse_loss(out[0].reshape(-1), z)
    policy_term = F.cross_entropy(out[1].reshape(-1, POLICY_SIZE), pi)
    total = value_term + policy_term
    aux_term = None
    if len(out) > 2:
        aux_term = F.mse_loss(out[2], compute_aux_targets(inputs[0]))
        total = total + aux_weight * aux_term
    return total, value_term, policy_term, aux_term


def train_network_pytorch(config, model, set, data_number=None, device=None):
    """`set`: the reference's list of 13-field records, or the 13 feature arrays of records.PackedSet.features()
    (a tuple) -- the same tensors without the detour through nested Python lists"""
    device = device or next(model.parameters()).device
    columns = _Columns(set)
    is_aux = isinstance(config.model_config, nets.AuxBaseResNetConfig)
    aux_weight = config.model_config.aux_weight if is_aux else 0.0
    optimizer = torch.optim.AdamW(model.parameters(), lr=config.learning_rate, weight_decay=config.weight_decay)
    means = _Means(["loss", "value_loss", "policy_loss"] + (["aux_loss"] if is_aux else []))
    model.train()
    for _ in range(config.epochs):
        for batch in columns.batches(config):
            *inputs, z, pi = (t.to(device) for t in batch)
            total, value_term, policy_term, aux_term = _objective(model, inputs, z.to(torch.float), pi.reshape(-1, POLICY_SIZE), aux_weight)
            total.backward()
            optimizer.step()
            optimizer.zero_grad()
            terms = {"loss": total, "value_loss": value_term, "policy_loss": policy_term}
            if is_aux:
                terms["aux_loss"] = aux_term
            means.add(**terms)
    model.eval()
    model._folded = None                      # the folded inference weights are stale now
    entry = {"model_version": config.model_version,
             "data_number": data_number if data_number is not None else selfplay.highest_data_number(config)}
    entry.update({name: means.mean(name) for name in ("loss", "value_loss", "policy_loss")})
    entry["backend"] = "pytorch"
    if is_aux:
        entry["aux_loss"] = means.mean("aux_loss")
    logs = selfplay.storage_root() / "logs"
    logs.mkdir(parents=True, exist_ok=True)
    with open(logs / "training_log.jsonl", "a") as f:
        f.write(json.dumps(entry) + "\n")
    return entry


def get_data_filenames(config):
    """ai.py:1943-1973: the newest `sets_to_train_with` set files"""
    path, newest = config.data_dir, selfplay.highest_data_number(config)
    listed = os.listdir(path) if os.path.isdir(path) else []
    have = {n.split(".")[0] for n in listed}
    if os.path.isdir(packed_dir(config)):                        # sets written with write_json=False exist only there
        listed += ["%s.txt" % n.split(".")[0] for n in sorted(os.listdir(packed_dir(config))) if n.split(".")[0] not in have]
    names = [n for n in listed if n.split(".")[0].isdigit() and int(n.split(".")[0]) > newest - config.sets_to_train_with]
    if config.shuffle:
        random.shuffle(names)
    return names


def packed_dir(config):
    """where make_training_set puts the packed form of its set files (outside data_dir: the reference lists that)"""
    return config.data_dir.rstrip("/") + "_packed"


def _load_sets(config, names):
    """every set of `names` in its cheapest available form: ("packed", records.PackedSet) when the packed file exists,
    else ("json", list of 13-field records) from the reference-format file (sets written by the reference itself, or
    before this build kept packed copies)"""
    from . import records
    out = []
    for name in names:
        path = "%s/%s.npz" % (packed_dir(config), name.split(".")[0])
        if os.path.isfile(path):
            out.append(("packed", records.PackedSet.load(path)))
        else:
            with open("%s/%s" % (config.data_dir, name)) as f:
                out.append(("json", json.load(f)))
    return out


_FIELD_DTYPES = [np.float32, np.float32, np.int64, np.int64, np.int64, np.float32, np.float32, np.int64, np.int64, np.int64,
                 np.int64, np.float64, np.float32]


def _features_of_records(recs):
    """the 13 feature arrays (dtypes of records.PackedSet.features) of reference-format records"""
    if not recs:
        return None
    return [np.array([r[f] for r in recs], dtype=_FIELD_DTYPES[f]) for f in range(13)]


def load_data_and_train_model(config, model, data=None):
    """ai.py:1871-1921, 'merge' style: newer sets are sampled fully, older ones with decay_factor ** age.
    Sets that have a packed copy are never materialised as lists: the same random.sample / random.shuffle draws are made
    over record indices (random.sample draws positions, so sampling range(n) consumes the generator exactly as sampling
    the n records does) and the feature arrays are built for the chosen records; sets that exist only in the reference's
    JSON format are read as the reference reads them -- set by set, so the two kinds mix."""
    if config.data_loading_style != "merge":
        raise NotImplementedError
    if data is None:
        names = sorted(get_data_filenames(config), key=lambda x: int(x.split(".")[0]), reverse=True)
        sets = _load_sets(config, names)
        if all(kind == "json" for kind, _ in sets):              # the reference's own path, statement for statement
            data = []
            for age, (_, s) in enumerate(sets):
                data.extend(random.sample(s, int(len(s) * config.decay_factor ** age)))
        else:
            picks = []                                           # (set index, record index) in the reference's order
            for age, (_, s) in enumerate(sets):
                n = len(s)
                picks.extend((age, j) for j in random.sample(range(n), int(n * config.decay_factor ** age)))
            if config.shuffle:
                random.shuffle(picks)
            parts = {}
            for a, (kind, s) in enumerate(sets):
                idx = [j for aa, j in picks if aa == a]
                if not idx:
                    continue
                parts[a] = s.features(np.array(idx, dtype=np.int64)) if kind == "packed" else \
                    _features_of_records([s[j] for j in idx])
            where, counters = np.zeros((len(picks), 2), dtype=np.int64), {a: 0 for a in parts}
            for k, (a, _) in enumerate(picks):
                where[k] = (a, counters[a]); counters[a] += 1
            first = parts[next(iter(parts))]
            feats = []
            for f in range(13):
                arr = np.empty((len(picks),) + first[f].shape[1:], dtype=first[f].dtype)
                for a in parts:
                    sel = where[:, 0] == a
                    arr[sel] = parts[a][f][where[sel, 1]]
                feats.append(arr)
            return train_network_pytorch(config, model, tuple(feats))
    else:
        data = [x for s in data for x in s]
    if config.shuffle:
        random.shuffle(data)
    return train_network_pytorch(config, model, data)
