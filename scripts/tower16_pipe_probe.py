"""probe: k_trunk16p (tile outermost, split-and-write under the matrix instructions) against k_trunk16s: accuracy against the
reference's logits (fixtures) and time by batch size"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import tetris_rl_amd as trl
from test_net_parity import load_net, packed_features, load_net64
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "net.npz"))
net = load_net(d, "b10f16", 10, 16, device="cuda")
grids, vec = packed_features(d["states"])
tg, tv = torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda()
d64 = np.load(os.path.join(G, "net64.npz"))
g64, v64 = packed_features(d64["states"])
for mode in ("split", "split_k", "fp32"):
    net.tower_mode = mode
    v, lg = net.heads(tg, tv)
    e8 = np.abs(lg.cpu().numpy() - d["b10f16/logits"]).max()
    v, lg = net.heads(torch.from_numpy(g64).cuda(), torch.from_numpy(v64).cuda())
    e64 = np.abs(lg.cpu().numpy()[:, d64["sampled"]] - d64["b10f16/sampled_logits"]).max()
    print("%-8s max logit error: 8 positions %.2e, 64 positions %.2e; range guard %d" % (mode, e8, e64, trl.engine.tower_overflow(0)), flush=True)
torch.manual_seed(0)
for m in (128, 256, 512, 1024, 2048):
    boards = (torch.rand((m, 1, 40, 10), device="cuda") < 0.3).float()
    line = []
    for mode in ("split", "split_k"):
        net.tower_mode = mode
        net._trunk_hip(boards, net._folded["packed_tower"]); torch.cuda.synchronize()
        ts = []
        for rep in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): net._trunk_hip(boards, net._folded["packed_tower"])
            e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        line.append("%s %.1f us" % ("k_trunk16p" if mode == "split" else "k_trunk16s", 1e3 * sorted(ts)[3]))
    print("%5d boards: %s" % (m, ", ".join(line)), flush=True)
