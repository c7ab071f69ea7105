"""k_trunk_wide_s (split-f16 wide towers) against k_trunk_wide (fp32 MFMA): accuracy against the reference's logits and time"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import tetris_rl_amd as trl
from test_net_parity import load_net, packed_features, load_formula_net, load_base32
G = os.path.join(ROOT, "tests", "golden")
d = np.load(os.path.join(G, "net.npz")); dd = np.load(os.path.join(G, "net_deep.npz")); dw = np.load(os.path.join(G, "net_wide.npz"))
grids, vec = packed_features(d["states"])
g, v = torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda()
cases = [("b20f64", load_formula_net(20, 64, 2064, device="cuda"), dd["b20f64/logits"]),
         ("b3f64", load_net(dw, "b3f64", 3, 64, device="cuda"), dw["b3f64/logits"]),
         ("b2f32", load_net(dw, "b2f32", 2, 32, device="cuda"), dw["b2f32/logits"])]
for tag in ("base32", "aux32"):
    dbase, nb = load_base32(G, tag, device="cuda")
    cases.append((tag, nb, dbase["%s/logits" % tag]))
for tag, net, ref in cases:
    for mode in ("fp32", "split"):
        net.tower_mode = mode
        val, logits = net.heads(g, v)
        print("%-7s %-6s max |logit - reference| = %.3e" % (tag, mode, np.abs(logits.cpu().numpy() - ref).max()), flush=True)
boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
net = cases[0][1]
for mode in ("fp32", "split"):
    net.tower_mode = mode
    line = []
    for n in (256, 512, 1024, 2048):
        b = boards[:n].contiguous()
        net.trunk(b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            o = net.trunk(b)
        e1.record(); e1.synchronize()
        line.append("%d: %.2f ms" % (n, e0.elapsed_time(e1) / 3))
    print("b20f64 %-6s per launch by boards  %s" % (mode, "  ".join(line)), flush=True)
