"""Policy/value net of the rollout path.

`AlphaSame` carries the reference's parameter names (architectures.py:60-142, state_dict keys in
SURVEY Appendix C) so `{n}.pt` files written by the reference load unchanged, but it is an
inference module with its own forward:

  * input is the packed feature layout the search kernel writes (trl_mcts_select): `grids`
    float[2][n][40*10] (side to move first, then opponent) and `vec` float[n][105];
  * both boards of all n leaves go through the residual tower as ONE batch of 2n images;
  * eval-mode batch norms are folded at `prepare()` time: the norm that follows a convolution
    directly (conv_block2's) into that convolution's weights, the others into per-channel
    scale/shift pairs applied with the ReLU; dense + BatchNorm1d pairs into one affine map;
  * output is what evaluate() returns (ai.py:1222-1319): sigmoid value [n] and softmaxed policy
    [n, 11583].
"""
import os

import torch
from torch import nn
import torch.nn.functional as F

ROWS, COLS = 40, 10
POLICY_SIZE = 27 * 39 * 11
PIECES_PER_PLAYER = 7 * 7
SCALARS_PER_PLAYER = 3


class AlphaSameConfig:
    """architectures.py:8-17 (only the fields the PyTorch net reads)"""

    def __init__(self, blocks=10, pooling_blocks=2, filters=16, cpool=4, dropout=0.25, kernels=1,
                 o_side_neurons=16, value_head_neurons=16):
        self.blocks, self.pooling_blocks, self.filters, self.cpool = blocks, pooling_blocks, filters, cpool
        self.dropout, self.kernels = dropout, kernels
        self.o_side_neurons, self.value_head_neurons = o_side_neurons, value_head_neurons


def _bn2d(c):
    return nn.BatchNorm2d(c)


class _Block(nn.Module):
    """parameter holder for one residual block: conv_block1 = {0: BN, 2: conv}, conv_block2 = {0: BN, 3: conv}"""

    def __init__(self, f):
        super().__init__()
        self.conv_block1 = nn.ModuleDict({"0": _bn2d(f), "2": nn.Conv2d(f, f, 3, padding=1, bias=False)})
        self.conv_block2 = nn.ModuleDict({"0": _bn2d(f), "3": nn.Conv2d(f, f, 3, padding=1, bias=False)})


def _affine(bn):
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return s, bn.bias - bn.running_mean * s


def _pack_conv3x3(w, f):
    """w [f out, f in, 3, 3] -> the B operands of v_mfma_f32_16x16x4_f32 in the order csrc/trl_net.hip reads them:
    [channel group][tap][j][N-tile][lane] = W[16 ntile + lane % 16][16 cg + 4 (lane // 16) + j][tap]"""
    G = f // 16
    lane = torch.arange(64, device=w.device)
    co, g = lane & 15, lane >> 4
    w = w.reshape(f, f, 9)
    out = torch.empty((G, 9, 4, G, 64), dtype=torch.float32, device=w.device)
    for cg in range(G):
        for nt in range(G):
            for j in range(4):
                out[cg, :, j, nt, :] = w[16 * nt + co, 16 * cg + 4 * g + j, :].transpose(0, 1)
    return out.reshape(-1)


SPLIT_SCALE = 2048.0        # csrc/trl_net.hip k_trunk16s: a = hi + lo / 2^11


class SplitRangeError(ValueError):
    """a folded conv weight is too large for the split towers (they form 2^11 * hi in f16: |w| must stay below 32)"""


def _check_split_range(w9):
    if not bool((w9.abs() < 31.0).all()):
        raise SplitRangeError("split tower: a folded conv weight of magnitude >= 31 does not fit the kernels' scaled f16 operand")


def _pack_conv3x3_split(w):
    """w [16 out, 16 in, 3, 3] -> the B operands of v_mfma_f32_16x16x32_f16 for the split tower (k_trunk16s):
    [K-step c = 0..4][hi | lo][lane][8 halves]; lane (n = lane % 16, g = lane // 16) holds W[n][8 (g & 1) + j][tap] for
    tap = 2 c + (g >> 1) (zero for the tenth half-step); hi = f16(w), lo = f16((w - hi) * 2^11).  Returned as float32
    words (bit patterns), 2560 per conv."""
    dev = w.device
    lane = torch.arange(64, device=dev)
    n, g = lane & 15, lane >> 4
    w9 = w.reshape(16, 16, 9).to(torch.float32)
    _check_split_range(w9)
    out = torch.zeros((5, 2, 64, 8), dtype=torch.float16, device=dev)
    for c in range(5):
        tap = 2 * c + (g >> 1)
        ok = tap <= 8
        tapc = torch.clamp(tap, max=8)
        for j in range(8):
            v = torch.where(ok, w9[n, 8 * (g & 1) + j, tapc], torch.zeros(64, device=dev))
            hi = v.to(torch.float16)
            out[c, 0, :, j] = hi
            out[c, 1, :, j] = ((v - hi.to(torch.float32)) * SPLIT_SCALE).to(torch.float16)
    return out.reshape(-1).view(torch.float32)


def _pack_conv3x3_split_wide(w, f):
    """w [f out, f in, 3, 3] (f = 32 or 64) -> the weight operands of k_trunk_wide_s:
    [K-step = kg * 9 + tap][hi | lo][channel group][lane][8 halves]; lane (n = lane % 16, g = lane // 16) of group nt holds
    W[16 nt + n][32 kg + 8 g + j][tap]; hi = f16(w), lo = f16((w - hi) * 2^11).  float32 words (bit patterns)."""
    dev = w.device
    nn_, kgs = f // 16, f // 32
    lane = torch.arange(64, device=dev)
    n, g = lane & 15, lane >> 4
    w9 = w.reshape(f, f, 9).to(torch.float32)
    _check_split_range(w9)
    v = torch.empty((kgs, 9, nn_, 64, 8), dtype=torch.float32, device=dev)
    for kg in range(kgs):
        for nt in range(nn_):
            for j in range(8):
                v[kg, :, nt, :, j] = w9[16 * nt + n, 32 * kg + 8 * g + j, :].transpose(0, 1)
    hi = v.to(torch.float16)
    lo = ((v - hi.to(torch.float32)) * SPLIT_SCALE).to(torch.float16)
    out = torch.stack([hi, lo], dim=2)                    # [kg][tap][2][nt][lane][8]
    return out.reshape(-1).view(torch.float32)


def _split_or_none(pack):
    """the split towers' packed weights, or None (= the fp32-MFMA tower kernels run instead) when a weight is out of
    their range -- said aloud: the two towers differ in speed, not in results"""
    try:
        return pack()
    except SplitRangeError as e:
        import warnings
        warnings.warn("%s; using the fp32 tower kernels for this network" % e, RuntimeWarning, stacklevel=3)
        return None


class SplitActivationRange(RuntimeWarning):
    """an activation of a split tower reached the f16 range (|a| >= 65 504): the result was recomputed on the fp32 kernels"""


def _split_overflowed(device):
    """eager callers of a split tower: wait for it and ask the library whether an activation left the f16 range
    (trl_tower_overflow); never inside a stream capture (SelfPlay / Battle ask at their own synchronisation points)"""
    if torch.cuda.is_current_stream_capturing():
        return False
    from . import engine as _eng
    torch.cuda.current_stream(device).synchronize()
    return _eng.tower_overflow(device.index or 0, reset=True) > 0


def _warn_split_range(what):
    import warnings
    warnings.warn("%s: an activation reached the f16 range of the split tower kernels (|a| >= 65504); this network runs on "
                  "the fp32 tower kernels from here on (tower_mode = 'fp32')" % what, SplitActivationRange, stacklevel=3)


class AlphaSame(nn.Module):
    def __init__(self, model_config=None, use_tanh=False):
        super().__init__()
        mc = model_config or AlphaSameConfig()
        if mc.kernels != 1:
            raise ValueError("only kernels=1 is supported (the reference default)")
        self.model_config, self.use_tanh = mc, use_tanh
        f = mc.filters
        self.conv1 = nn.Conv2d(1, f, 5, padding=2, bias=False)
        self.res_blocks = nn.ModuleList([_Block(f) for _ in range(mc.blocks)])
        self.batchnorm1 = _bn2d(f)
        self.kernel1 = nn.Conv2d(f, mc.kernels, 1, bias=False)
        self.batchnorm2 = _bn2d(mc.kernels)
        grid_out = ROWS * COLS * mc.kernels
        self.osidedense = nn.ModuleDict({"0": nn.Linear(grid_out, mc.o_side_neurons), "1": nn.BatchNorm1d(mc.o_side_neurons)})
        head_in = grid_out + mc.o_side_neurons + 2 * PIECES_PER_PLAYER + 2 * SCALARS_PER_PLAYER + 1      # 521
        self.policy_head = nn.Linear(head_in, POLICY_SIZE)
        self.value_head = nn.ModuleDict({"0": nn.Linear(head_in, mc.value_head_neurons),
                                         "1": nn.BatchNorm1d(mc.value_head_neurons),
                                         "3": nn.Linear(mc.value_head_neurons, 1)})
        self._folded = None
        self.use_hip_trunk = True       # fused HIP tower when the net is on the GPU and filters is 16, 32 or 64
        # "split" = k_trunk16s / k_trunk_wide_s (f16 matrix rate, operands split in two halves, fp32 accuracy),
        # "fp32" = k_trunk16 / k_trunk_wide (fp32 matrix instructions)
        self.tower_mode = os.environ.get("TRL_TOWER", "split")
        self.eval()

    # -- folding ------------------------------------------------------------------------------
    @torch.no_grad()
    def prepare(self):
        """fold the eval-mode batch norms; call after loading weights / moving devices"""
        fd = {"blocks": []}
        for blk in self.res_blocks:
            s1, t1 = _affine(blk.conv_block1["0"])
            s2, t2 = _affine(blk.conv_block2["0"])
            wa = blk.conv_block1["2"].weight * s2.view(-1, 1, 1, 1)
            fd["blocks"].append((s1.view(1, -1, 1, 1).contiguous(), t1.view(1, -1, 1, 1).contiguous(),
                                 wa.contiguous(), t2.contiguous(), blk.conv_block2["3"].weight.contiguous()))
        s, t = _affine(self.batchnorm1)
        fd["tail"] = (s.view(1, -1, 1, 1).contiguous(), t.view(1, -1, 1, 1).contiguous())
        sk, tk = _affine(self.batchnorm2)
        fd["k1"] = ((self.kernel1.weight * sk.view(-1, 1, 1, 1)).contiguous(), tk.contiguous())
        so, to = _affine(self.osidedense["1"])
        fd["oside"] = ((self.osidedense["0"].weight * so.view(-1, 1)).contiguous(), (self.osidedense["0"].bias * so + to).contiguous())
        sv, tv = _affine(self.value_head["1"])
        fd["v0"] = ((self.value_head["0"].weight * sv.view(-1, 1)).contiguous(), (self.value_head["0"].bias * sv + tv).contiguous())
        fd["packed_tower"] = self._pack_tower(fd) if (self.model_config.filters in self.HIP_TOWER_FILTERS and self.model_config.kernels == 1
                                              and self.conv1.weight.is_cuda) else None
        # the 16-filter tower on the f16 matrix rate with split operands (k_trunk16s): fp32 accuracy, see csrc/trl_net.hip
        fd["packed_split"] = _split_or_none(lambda: self._pack_tower_split(fd)) if fd["packed_tower"] is not None else None
        self._folded = fd
        return self

    HIP_TOWER_FILTERS = (16, 32, 64)

    def _pack_tower(self, fd):
        """weights in the layout of csrc/trl_net.hip (B operands of v_mfma_f32_16x16x4_f32); filters 16, 32 or 64:
        a 3x3 conv becomes [channel group][tap][j][N-tile][lane] = W[16 ntile + lane % 16][16 cg + 4 (lane // 16) + j][tap]"""
        dev = self.conv1.weight.device
        f = self.model_config.filters

        def pack_conv(w):
            return _pack_conv3x3(w, f)

        parts = [self.conv1.weight.reshape(f, 25).reshape(-1)]
        for s1, t1, wa, ba, wb in fd["blocks"]:
            parts += [s1.reshape(-1), t1.reshape(-1), pack_conv(wa), ba.reshape(-1), pack_conv(wb)]
        s, t = fd["tail"]
        wk, bk = fd["k1"]
        # padding: the wide kernels prefetch the weights one K-step past the end of a conv
        parts += [s.reshape(-1), t.reshape(-1), wk.reshape(-1), bk.reshape(-1), torch.zeros(15 + (0 if f == 16 else 4096), device=dev)]
        return torch.cat([p.to(torch.float32) for p in parts]).contiguous()

    def _pack_tower_split(self, fd):
        """packed weights of k_trunk16s: stem [16][25] fp32; per block s1[16] t1[16] convA (2560 words) ba[16] convB (2560);
        tail s[16] t[16] wk[16] bk + padding"""
        dev = self.conv1.weight.device
        f = self.model_config.filters
        pack = _pack_conv3x3_split if f == 16 else (lambda w: _pack_conv3x3_split_wide(w, f))
        parts = [self.conv1.weight.reshape(f, 25).reshape(-1)]
        for s1, t1, wa, ba, wb in fd["blocks"]:
            parts += [s1.reshape(-1), t1.reshape(-1), pack(wa), ba.reshape(-1), pack(wb)]
        s, t = fd["tail"]
        wk, bk = fd["k1"]
        # padding: the wide kernel prefetches the weights one K-step past the end of a conv
        parts += [s.reshape(-1), t.reshape(-1), wk.reshape(-1), bk.reshape(-1), torch.zeros(15 + (0 if f == 16 else 4096), device=dev)]
        return torch.cat([p.to(torch.float32) if p.dtype != torch.float32 else p for p in parts]).contiguous()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._folded = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._folded = None
        return r

    # -- inference ----------------------------------------------------------------------------
    @torch.no_grad()
    def trunk(self, boards):
        """boards float[m, 1, 40, 10] -> float[m, 400] (process_grid of the reference, architectures.py:121-127)"""
        fd = self._folded or self.prepare()._folded
        if fd["packed_tower"] is not None and self.use_hip_trunk:
            out = self._trunk_hip(boards, fd["packed_tower"])
            if self.tower_mode == "split" and fd.get("packed_split") is not None and _split_overflowed(boards.device):
                _warn_split_range("AlphaSame.trunk")
                self.tower_mode = "fp32"
                out = self._trunk_hip(boards, fd["packed_tower"])
            return out
        x = F.conv2d(boards, self.conv1.weight, padding=2)
        for s1, t1, wa, ba, wb in fd["blocks"]:
            h = torch.relu(x * s1 + t1)
            h = torch.relu(F.conv2d(h, wa, ba, padding=1))
            x = x + F.conv2d(h, wb, padding=1)
        s, t = fd["tail"]
        wk, bk = fd["k1"]
        return torch.relu(F.conv2d(torch.relu(x * s + t), wk, bk)).flatten(1)

    def _trunk_hip(self, boards, packed, need=None, out=None):
        """fused HIP tower (csrc/trl_net.hip) on the current stream; raises if the library is missing.
        need: optional uint8 [m] -- boards with 0 are skipped (their output rows stay zero, or, with a caller's
        `out` buffer, untouched)"""
        from . import engine as _eng
        L = _eng.load_library()
        m = boards.shape[0]
        x = boards.reshape(m, ROWS * COLS).contiguous()
        if out is not None:
            pass
        elif need is None:
            out = torch.empty((m, ROWS * COLS), dtype=torch.float32, device=boards.device)
        else:
            out = torch.zeros((m, ROWS * COLS), dtype=torch.float32, device=boards.device)
        split = self._folded.get("packed_split") if (self._folded and self.tower_mode == "split") else None
        if split is not None and packed is self._folded["packed_tower"] and self.model_config.filters == 16:
            rc = L.trl_alphasame_trunk16_split(x.data_ptr(), m, split.data_ptr(), len(self.res_blocks), out.data_ptr(),
                                               None if need is None else need.data_ptr(),
                                               torch.cuda.current_stream(boards.device).cuda_stream)
        elif split is not None and packed is self._folded["packed_tower"]:
            rc = L.trl_trunk_wide_split(x.data_ptr(), m, split.data_ptr(), self.model_config.filters, len(self.res_blocks), 0, 1,
                                        out.data_ptr(), None if need is None else need.data_ptr(),
                                        torch.cuda.current_stream(boards.device).cuda_stream)
        else:
            rc = L.trl_alphasame_trunk(x.data_ptr(), m, packed.data_ptr(), self.model_config.filters, len(self.res_blocks),
                                       out.data_ptr(), None if need is None else need.data_ptr(),
                                       torch.cuda.current_stream(boards.device).cuda_stream)
        if rc != 0:
            raise _eng.EngineError("trl_alphasame_trunk failed: %s" % L.trl_last_error().decode())
        return out

    @property
    def supports_tower_cache(self):
        """True when the tower runs as the fused HIP kernel, which can skip boards (trl_mcts_zcache flow)"""
        fd = self._folded or self.prepare()._folded
        return fd["packed_tower"] is not None and self.use_hip_trunk

    tower_floats_per_board = ROWS * COLS

    def device_heads(self):
        """trl_alphasame_heads for trl_mcts_expand_alphasame (pointers into the folded tensors of prepare())"""
        from . import engine as _eng
        fd = self._folded or self.prepare()._folded
        mc = self.model_config
        if mc.o_side_neurons != 16 or mc.value_head_neurons != 16:
            return None
        (wo, bo), (wv, bv) = fd["oside"], fd["v0"]
        w3, b3 = self.value_head["3"].weight.reshape(-1).contiguous(), float(self.value_head["3"].bias.item())
        fd["_keep"] = (w3,)
        ph = self.policy_head
        return _eng.TrlAlphaSameHeads(wo.data_ptr(), bo.data_ptr(), wv.data_ptr(), bv.data_ptr(), w3.data_ptr(),
                                      ph.weight.data_ptr(), ph.bias.data_ptr(), b3, int(bool(self.use_tanh)))

    @torch.no_grad()
    def tower_need(self, grids, need, out=None):
        """tower outputs [2n, 400] of the boards flagged in need (uint8 [2n]); the others are zero (left as they
        are in a caller-provided `out`)"""
        fd = self._folded or self.prepare()._folded
        return self._trunk_hip(grids.reshape(-1, 1, ROWS, COLS), fd["packed_tower"], need, out)

    @property
    def supports_tower_legal(self):
        """True when the tick's tower launch can carry the search's legal-move job (trl_mcts_tower16_legal): the split
        16-filter tower"""
        fd = self._folded or self.prepare()._folded
        return bool(self.tower_mode == "split" and fd.get("packed_split") is not None and self.model_config.filters == 16
                    and self.use_hip_trunk)

    def tower_legal(self, engine, grids, need, out):
        """tower_need(grids, need, out) on the engine's stream with the legal-move search of the engine's pending leaves
        in the same launch (Engine.mcts_legal_in_tower)"""
        fd = self._folded or self.prepare()._folded
        m = grids.shape[0] * grids.shape[1]
        engine.mcts_tower16_legal(grids.data_ptr(), m, fd["packed_split"].data_ptr(), len(self.res_blocks), out.data_ptr(),
                                  need.data_ptr())
        return out

    @torch.no_grad()
    def head_inputs(self, grids, vec, z=None):
        """-> (value [n] before the output activation, x [n, 521] = input of the policy head);
        z: precomputed tower outputs [2n, 400] (side to move first), else the tower runs on grids"""
        fd = self._folded or self.prepare()._folded
        n = vec.shape[0]
        if z is None:
            z = self.trunk(grids.reshape(2 * n, 1, ROWS, COLS))
        wo, bo = fd["oside"]
        o = torch.relu(F.linear(z[n:], wo, bo))
        # concat order of the reference (architectures.py:132-140)
        x = torch.cat([z[:n], vec[:, :52], o, vec[:, 52:]], dim=1).contiguous()
        wv, bv = fd["v0"]
        v = F.linear(torch.relu(F.linear(x, wv, bv)), self.value_head["3"].weight, self.value_head["3"].bias).reshape(n)
        return v, x

    @torch.no_grad()
    def heads(self, grids, vec):
        """-> (value_pre_activation [n], policy logits [n, 11583])"""
        v, x = self.head_inputs(grids, vec)
        return v, F.linear(x, self.policy_head.weight, self.policy_head.bias)

    @torch.no_grad()
    def evaluate_head_inputs(self, grids, vec, z=None):
        """(value [n], x [n, 521]) for trl_mcts_expand_head, which forms only the legal logits itself"""
        v, x = self.head_inputs(grids, vec, z)
        return (torch.tanh(v) if self.use_tanh else torch.sigmoid(v)), x

    @torch.no_grad()
    def evaluate_packed(self, grids, vec):
        """(value [n], softmaxed policy [n, 11583]) -- evaluate() of the reference on a batch"""
        v, logits = self.heads(grids, vec)
        v = torch.tanh(v) if self.use_tanh else torch.sigmoid(v)
        return v, torch.softmax(logits, dim=1)

    def forward(self, a_grid, a_pieces, a_b2b, a_combo, a_garbage, o_grid, o_pieces, o_b2b, o_combo, o_garbage, color):
        """the reference's 11-tensor signature (architectures.py:119-142), batch-first"""
        n = a_grid.shape[0]
        grids = torch.stack([a_grid.reshape(n, -1), o_grid.reshape(n, -1)]).to(torch.float32)
        cols = [a_pieces.reshape(n, -1), a_b2b.reshape(n, 1), a_combo.reshape(n, 1), a_garbage.reshape(n, 1),
                o_pieces.reshape(n, -1), o_b2b.reshape(n, 1), o_combo.reshape(n, 1), o_garbage.reshape(n, 1),
                color.reshape(n, 1)]
        vec = torch.cat([c.to(torch.float32) for c in cols], dim=1)
        v, logits = self.heads(grids, vec)
        v = torch.tanh(v) if self.use_tanh else torch.sigmoid(v)
        return v.reshape(n, 1), logits


def model_for_config(model_config=None, use_tanh=False):
    """the net class the reference instantiates for a model_config (ai.py:1048-1053): AuxBaseResNetConfig ->
    AuxBaseResNet, BaseResNetConfig -> BaseResNet, anything else (AlphaSameConfig) -> AlphaSame"""
    if isinstance(model_config, AuxBaseResNetConfig):
        return AuxBaseResNet(model_config, use_tanh)
    if isinstance(model_config, BaseResNetConfig):
        return BaseResNet(model_config, use_tanh)
    return AlphaSame(model_config, use_tanh)


def load_reference_checkpoint(path, model_config=None, device="cpu"):
    """`torch.save(model.state_dict(), '{n}.pt')` files of the reference (ai.py:1082, 2192)"""
    net = model_for_config(model_config)
    sd = torch.load(path, map_location="cpu", weights_only=True)
    net.load_state_dict(sd)
    return net.to(device).prepare()


# ---- the reference's current default family (architectures.py:150-353) -------------------------------------
class BaseResNetConfig:
    def __init__(self, blocks=8, filters=32, opp_hidden=128, own_kernels=4, value_hidden=16):
        self.blocks, self.filters, self.opp_hidden = blocks, filters, opp_hidden
        self.own_kernels, self.value_hidden = own_kernels, value_hidden


class AuxBaseResNetConfig(BaseResNetConfig):
    def __init__(self, aux_hidden=16, aux_weight=1.5, **kw):
        super().__init__(**kw)
        self.aux_hidden, self.aux_weight = aux_hidden, aux_weight


class _PostBlock(nn.Module):
    """parameter holder: conv1, bn1, conv2, bn2 (post-activation block, architectures.py:159-170)"""

    def __init__(self, f):
        super().__init__()
        self.conv1 = nn.Conv2d(f, f, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(f)
        self.conv2 = nn.Conv2d(f, f, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(f)


def _fold_conv_bn(conv, bn):
    s, t = _affine(bn)
    return (conv.weight * s.view(-1, 1, 1, 1)).contiguous(), t.contiguous()


def _fold_lin_bn(lin, bn):
    s, t = _affine(bn)
    return (lin.weight * s.view(-1, 1)).contiguous(), (lin.bias * s + t).contiguous()


class BaseResNet(nn.Module):
    """Inference module with the reference's parameter names (stem.0/.1, trunk.i.{conv1,bn1,conv2,bn2},
    opp_collapse, opp_encode, bias_project, own_collapse, policy_head, value_head) and the same packed
    feature interface as AlphaSame.  Every conv is followed directly by its batch norm, so all norms fold
    into the convolutions; on the GPU with 32 or 64 filters the tower runs as the fused HIP kernel
    (trl_resnet_trunk, csrc/trl_net.hip), otherwise as PyTorch convolutions."""

    has_aux = False
    HIP_TOWER_FILTERS = (32, 64)

    def __init__(self, model_config=None, use_tanh=False):
        super().__init__()
        mc = model_config or BaseResNetConfig()
        self.model_config, self.use_tanh = mc, use_tanh
        f = mc.filters
        self.stem = nn.ModuleDict({"0": nn.Conv2d(1, f, 3, padding=1, bias=False), "1": nn.BatchNorm2d(f)})
        self.trunk = nn.ModuleList([_PostBlock(f) for _ in range(mc.blocks)])
        self.opp_collapse = nn.ModuleDict({"0": nn.Conv2d(f, 1, 1, bias=False), "1": nn.BatchNorm2d(1)})
        opp_in = ROWS * COLS + PIECES_PER_PLAYER + SCALARS_PER_PLAYER
        self.opp_encode = nn.ModuleDict({"0": nn.Linear(opp_in, mc.opp_hidden), "1": nn.BatchNorm1d(mc.opp_hidden)})
        self.bias_project = nn.Linear(mc.opp_hidden + PIECES_PER_PLAYER + SCALARS_PER_PLAYER + 1, f)
        self.own_collapse = nn.ModuleDict({"0": nn.Conv2d(f, mc.own_kernels, 1, bias=False), "1": nn.BatchNorm2d(mc.own_kernels)})
        head_in = mc.own_kernels * ROWS * COLS + PIECES_PER_PLAYER + SCALARS_PER_PLAYER + 1
        self.policy_head = nn.Linear(head_in, POLICY_SIZE)
        self.value_head = nn.ModuleDict({"0": nn.Linear(head_in, mc.value_hidden), "1": nn.BatchNorm1d(mc.value_hidden),
                                         "3": nn.Linear(mc.value_hidden, 1)})
        self._folded = None
        self.use_hip_trunk = True       # fused HIP tower when the net is on the GPU and filters is 32 or 64
        self.tower_mode = os.environ.get("TRL_TOWER", "split")     # "split": k_trunk_wide_s, "fp32": k_trunk_wide
        self.eval()

    @torch.no_grad()
    def prepare(self):
        fd = {"stem": _fold_conv_bn(self.stem["0"], self.stem["1"]),
              "blocks": [(_fold_conv_bn(b.conv1, b.bn1), _fold_conv_bn(b.conv2, b.bn2)) for b in self.trunk],
              "opp": _fold_conv_bn(self.opp_collapse["0"], self.opp_collapse["1"]),
              "enc": _fold_lin_bn(self.opp_encode["0"], self.opp_encode["1"]),
              "own": _fold_conv_bn(self.own_collapse["0"], self.own_collapse["1"]),
              "v0": _fold_lin_bn(self.value_head["0"], self.value_head["1"])}
        if self.has_aux:
            fd["a0"] = _fold_lin_bn(self.aux_head["0"], self.aux_head["1"])
        mc = self.model_config
        fd["packed_tower"] = None
        if mc.filters in self.HIP_TOWER_FILTERS and mc.own_kernels + 1 <= 8 and self.stem["0"].weight.is_cuda:
            f = mc.filters
            parts = [fd["stem"][0].reshape(-1), fd["stem"][1]]
            for (w1, b1), (w2, b2) in fd["blocks"]:
                parts += [b1, _pack_conv3x3(w1, f), b2, _pack_conv3x3(w2, f)]
            # tail: the own_kernels rows of own_collapse, then the row of opp_collapse (kout = own_kernels + 1)
            parts += [fd["own"][0].reshape(-1), fd["opp"][0].reshape(-1), torch.zeros(4096, device=fd["own"][0].device)]
            fd["packed_tower"] = torch.cat([p.reshape(-1).to(torch.float32) for p in parts]).contiguous()
            def pack_split():
                parts = [fd["stem"][0].reshape(-1), fd["stem"][1]]
                for (w1, b1), (w2, b2) in fd["blocks"]:
                    parts += [b1, _pack_conv3x3_split_wide(w1, f), b2, _pack_conv3x3_split_wide(w2, f)]
                parts += [fd["own"][0].reshape(-1), fd["opp"][0].reshape(-1), torch.zeros(4096, device=fd["own"][0].device)]
                return torch.cat([p.reshape(-1).to(torch.float32) for p in parts]).contiguous()
            fd["packed_split"] = _split_or_none(pack_split)
        self._folded = fd
        return self

    def _tower_hip(self, boards, packed, need=None):
        """-> float[m, own_kernels + 1, 400]: raw 1x1 projections of the final feature map (trl_resnet_trunk);
        need: optional uint8 [m] -- boards with 0 are skipped (their rows are zero)"""
        from . import engine as _eng
        L = _eng.load_library()
        m = boards.shape[0]
        kout = self.model_config.own_kernels + 1
        x = boards.reshape(m, ROWS * COLS).contiguous()
        alloc = torch.empty if need is None else torch.zeros
        out = alloc((m, kout, ROWS * COLS), dtype=torch.float32, device=boards.device)
        split = self._folded.get("packed_split") if (self._folded and self.tower_mode == "split") else None
        if split is not None and packed is self._folded["packed_tower"]:
            rc = L.trl_trunk_wide_split(x.data_ptr(), m, split.data_ptr(), self.model_config.filters, len(self.trunk), 1, kout,
                                        out.data_ptr(), None if need is None else need.data_ptr(),
                                        torch.cuda.current_stream(boards.device).cuda_stream)
        else:
            rc = L.trl_resnet_trunk(x.data_ptr(), m, packed.data_ptr(), self.model_config.filters, len(self.trunk), kout,
                                    out.data_ptr(), None if need is None else need.data_ptr(),
                                    torch.cuda.current_stream(boards.device).cuda_stream)
        if rc != 0:
            raise _eng.EngineError("trl_resnet_trunk failed: %s" % L.trl_last_error().decode())
        return out

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._folded = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._folded = None
        return r

    @property
    def supports_tower_cache(self):
        """True when the tower runs as the fused HIP kernel (it can skip boards; trl_mcts_zcache flow)"""
        fd = self._folded or self.prepare()._folded
        return fd["packed_tower"] is not None and self.use_hip_trunk

    @property
    def tower_floats_per_board(self):
        return (self.model_config.own_kernels + 1) * ROWS * COLS

    @torch.no_grad()
    def tower_need(self, grids, need):
        """raw tower projections [2n, (own_kernels + 1) * 400] of the boards flagged in need (uint8 [2n])"""
        fd = self._folded or self.prepare()._folded
        n2 = grids.shape[0] * grids.shape[1]
        return self._tower_hip(grids.reshape(n2, ROWS * COLS), fd["packed_tower"], need).reshape(n2, -1)

    @torch.no_grad()
    def head_inputs(self, grids, vec, z=None):
        """z: precomputed tower projections [2n, (own_kernels + 1) * 400] (side to move first), else the tower runs"""
        fd = self._folded or self.prepare()._folded
        n = vec.shape[0]
        a_extra, o_extra, color = vec[:, :52], vec[:, 52:104], vec[:, 104:105]
        hip = z is not None or (fd["packed_tower"] is not None and self.use_hip_trunk)
        if hip:
            if z is None:
                z = self._tower_hip(grids.reshape(2 * n, ROWS * COLS), fd["packed_tower"])
                if self.tower_mode == "split" and fd.get("packed_split") is not None and _split_overflowed(grids.device):
                    _warn_split_range(type(self).__name__)
                    self.tower_mode = "fp32"
                    z = self._tower_hip(grids.reshape(2 * n, ROWS * COLS), fd["packed_tower"])
            z = z.reshape(2 * n, self.model_config.own_kernels + 1, ROWS * COLS)
            k = self.model_config.own_kernels
            o_col = torch.relu(z[n:, k] + fd["opp"][1])
        else:
            x = torch.relu(F.conv2d(grids.reshape(2 * n, 1, ROWS, COLS), *fd["stem"], padding=1))
            for (w1, b1), (w2, b2) in fd["blocks"]:
                h = torch.relu(F.conv2d(x, w1, b1, padding=1))
                x = torch.relu(F.conv2d(h, w2, b2, padding=1) + x)
            a_feat, o_feat = x[:n], x[n:]
            o_col = torch.relu(F.conv2d(o_feat, *fd["opp"])).flatten(1)
        opp_repr = torch.relu(F.linear(torch.cat([o_col, o_extra], dim=1), *fd["enc"]))
        bias_vec = F.linear(torch.cat([opp_repr, a_extra, color], dim=1), self.bias_project.weight, self.bias_project.bias)
        if hip:
            # own_collapse is linear: W (a_feat + bias) + t = [W a_feat from the kernel] + W bias + t
            wo, to = fd["own"]
            flat = torch.relu(z[:n, :k] + (F.linear(bias_vec, wo.reshape(k, -1)) + to).unsqueeze(2)).flatten(1)
        else:
            flat = torch.relu(F.conv2d(a_feat + bias_vec.view(n, -1, 1, 1), *fd["own"])).flatten(1)
        hx = torch.cat([flat, a_extra, color], dim=1).contiguous()
        v = F.linear(torch.relu(F.linear(hx, *fd["v0"])), self.value_head["3"].weight, self.value_head["3"].bias).reshape(n)
        return v, hx

    @torch.no_grad()
    def heads(self, grids, vec):
        v, hx = self.head_inputs(grids, vec)
        return v, F.linear(hx, self.policy_head.weight, self.policy_head.bias)

    @torch.no_grad()
    def evaluate_packed(self, grids, vec):
        v, logits = self.heads(grids, vec)
        return (torch.tanh(v) if self.use_tanh else torch.sigmoid(v)), torch.softmax(logits, dim=1)

    @torch.no_grad()
    def evaluate_head_inputs(self, grids, vec, z=None):
        v, hx = self.head_inputs(grids, vec, z)
        return (torch.tanh(v) if self.use_tanh else torch.sigmoid(v)), hx

    def _pack(self, a_grid, a_pieces, a_b2b, a_combo, a_garbage, o_grid, o_pieces, o_b2b, o_combo, o_garbage, color):
        n = a_grid.shape[0]
        grids = torch.stack([a_grid.reshape(n, -1), o_grid.reshape(n, -1)]).to(torch.float32)
        cols = [a_pieces.reshape(n, -1), a_b2b.reshape(n, 1), a_combo.reshape(n, 1), a_garbage.reshape(n, 1),
                o_pieces.reshape(n, -1), o_b2b.reshape(n, 1), o_combo.reshape(n, 1), o_garbage.reshape(n, 1), color.reshape(n, 1)]
        return grids, torch.cat([c.to(torch.float32) for c in cols], dim=1)

    @torch.no_grad()
    def forward(self, *features):
        grids, vec = self._pack(*features)
        v, logits = self.heads(grids, vec)
        v = torch.tanh(v) if self.use_tanh else torch.sigmoid(v)
        return v.reshape(-1, 1), logits


class AuxBaseResNet(BaseResNet):
    """BaseResNet + the 2-output auxiliary head (holes, height; architectures.py:308-353); forward returns
    (value, policy logits, aux) like the reference, the search ignores aux (ai.py:1316)."""

    has_aux = True

    def __init__(self, model_config=None, use_tanh=False):
        mc = model_config or AuxBaseResNetConfig()
        super().__init__(mc, use_tanh)
        head_in = mc.own_kernels * ROWS * COLS + PIECES_PER_PLAYER + SCALARS_PER_PLAYER + 1
        self.aux_head = nn.ModuleDict({"0": nn.Linear(head_in, mc.aux_hidden), "1": nn.BatchNorm1d(mc.aux_hidden),
                                       "3": nn.Linear(mc.aux_hidden, 2)})
        self.eval()

    @torch.no_grad()
    def forward(self, *features):
        grids, vec = self._pack(*features)
        fd = self._folded or self.prepare()._folded
        v, hx = self.head_inputs(grids, vec)
        logits = F.linear(hx, self.policy_head.weight, self.policy_head.bias)
        aux = torch.sigmoid(F.linear(torch.relu(F.linear(hx, *fd["a0"])), self.aux_head["3"].weight, self.aux_head["3"].bias))
        v = torch.tanh(v) if self.use_tanh else torch.sigmoid(v)
        return v.reshape(-1, 1), logits, aux


# ---- Keras -> PyTorch weight layout of the default family (util.py:2369-2464) ------------------------------------------
# The reference's current default configuration trains with Keras and stores `.keras` checkpoints; its
# port_keras_auxbaseresnet_to_pytorch names every weight-bearing Keras layer after the PyTorch state_dict prefix it
# feeds.  Keras is not importable in this build's environment, so the port takes the weights as plain arrays
# (layer name -> keras_layer.get_weights()) -- what `{l.name: l.get_weights() for l in keras_model.layers}` gives where
# Keras exists.  PARITY UNPINNED: no Keras output can be produced here to compare with; the layout rules below are the
# reference's (kernels HWIO -> OIHW, Dense (in, out) -> (out, in), BatchNormalization [gamma, beta, mean, var], and the
# NHWC flatten order of the own-side map in front of the three heads), checked by a synthetic forward-equivalence test.
def _keras_layer_table(blocks):
    """(keras layer name, state_dict prefix, kind) in the order of the reference's port"""
    t = [("stem_conv", "stem.0", "conv"), ("stem_bn", "stem.1", "bn")]
    for i in range(blocks):
        t += [("block%d_conv1" % i, "trunk.%d.conv1" % i, "conv"), ("block%d_bn1" % i, "trunk.%d.bn1" % i, "bn"),
              ("block%d_conv2" % i, "trunk.%d.conv2" % i, "conv"), ("block%d_bn2" % i, "trunk.%d.bn2" % i, "bn")]
    t += [("opp_collapse_conv", "opp_collapse.0", "conv"), ("opp_collapse_bn", "opp_collapse.1", "bn"),
          ("opp_encode_dense", "opp_encode.0", "dense"), ("opp_encode_bn", "opp_encode.1", "bn"),
          ("bias_project", "bias_project", "dense"),
          ("own_collapse_conv", "own_collapse.0", "conv"), ("own_collapse_bn", "own_collapse.1", "bn"),
          ("policy", "policy_head", "head"),
          ("value_hidden", "value_head.0", "head"), ("value_hidden_bn", "value_head.1", "bn"), ("value", "value_head.3", "dense"),
          ("aux_hidden", "aux_head.0", "head"), ("aux_hidden_bn", "aux_head.1", "bn"), ("aux", "aux_head.3", "dense")]
    return t


def port_keras_weights(keras_weights, model_config=None, use_tanh=False, aux=True):
    """keras_weights: {layer name: [arrays as keras_layer.get_weights() returns them]} of a Keras AuxBaseResNet (or
    BaseResNet with aux=False) built by the reference's gen_auxbaseresnet_keras -> the PyTorch-layout net of this package
    with those weights (eval mode, CPU).  Raises KeyError for a missing layer and ValueError for a shape that does not
    fit, naming the layer."""
    import numpy as np
    mc = model_config or (AuxBaseResNetConfig() if aux else BaseResNetConfig())
    net = AuxBaseResNet(mc, use_tanh) if aux else BaseResNet(mc, use_tanh)
    sd = net.state_dict()
    k = mc.own_kernels
    flat = k * ROWS * COLS

    def put(key, arr, layer):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        if tuple(sd[key].shape) != arr.shape:
            raise ValueError("Keras layer %r: %s has shape %s, the PyTorch parameter %s" % (layer, key, arr.shape, tuple(sd[key].shape)))
        sd[key] = torch.from_numpy(arr.copy())

    for layer, prefix, kind in _keras_layer_table(mc.blocks):
        if not aux and prefix.startswith("aux_head"):
            continue
        if layer not in keras_weights:
            raise KeyError("Keras layer %r (-> %s) is missing" % (layer, prefix))
        w = [np.asarray(a) for a in keras_weights[layer]]
        if kind == "conv":                                  # Conv2D kernel (kH, kW, in, out), no bias
            put(prefix + ".weight", w[0].transpose(3, 2, 0, 1), layer)
        elif kind == "bn":                                  # [gamma, beta, moving_mean, moving_variance]
            for name, a in zip(("weight", "bias", "running_mean", "running_var"), w):
                put("%s.%s" % (prefix, name), a, layer)
        else:                                               # Dense kernel (in, out), bias (out,)
            kern = w[0]
            if kind == "head":
                # Keras flattens the own-side map (B, ROWS, COLS, K) in [r, c, k] order, PyTorch's flatten(1) of
                # (B, K, ROWS, COLS) in [k, r, c] order: re-order the first K * ROWS * COLS input rows, keep the rest
                own = kern[:flat].reshape(ROWS, COLS, k, -1).transpose(2, 0, 1, 3).reshape(flat, -1)
                kern = np.concatenate([own, kern[flat:]], axis=0)
            put(prefix + ".weight", kern.T, layer)
            put(prefix + ".bias", w[1], layer)
    net.load_state_dict(sd)
    return net.eval()
