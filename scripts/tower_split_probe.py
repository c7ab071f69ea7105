"""k_trunk16s (split-f16 tower) against k_trunk16 (fp32 MFMA tower): accuracy against the reference's logits and time.
usage: python scripts/tower_split_probe.py   (TRL_SPLIT_ILV=0|1 forces the grouped / interleaved fragment-read form of the split kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import tetris_rl_amd as trl
from test_net_parity import load_net, packed_features
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "net.npz"))
net = load_net(d, "b10f16", 10, 16, device="cuda")
grids, vec = packed_features(d["states"])
g, v = torch.from_numpy(grids).cuda(), torch.from_numpy(vec).cuda()
print("TRL_SPLIT_ILV =", os.environ.get("TRL_SPLIT_ILV", "by board count (default)"))
for mode in ("fp32", "split"):
    net.tower_mode = mode
    val, logits = net.heads(g, v)
    print("%-6s max |logit - reference| = %.3e   value err %.2e" % (mode, np.abs(logits.cpu().numpy() - d["b10f16/logits"]).max(),
          np.abs(torch.sigmoid(val).cpu().numpy() - d["b10f16/values"]).max()), flush=True)
boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
outs = {}
for mode in ("fp32", "split"):
    net.tower_mode = mode
    line = []
    for n in (256, 512, 768, 1024, 1536, 2048):
        b = boards[:n].contiguous()
        net.trunk(b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o = net.trunk(b)
        e1.record(); e1.synchronize()
        line.append("%d: %.0f" % (n, e0.elapsed_time(e1) / 20 * 1e3))
    print("%-6s us per launch by boards  %s" % (mode, "  ".join(line)), flush=True)
    outs[mode] = o
print("split vs fp32 on random boards: max abs diff %.3e (max |out| %.2f)" % (float((outs["split"] - outs["fp32"]).abs().max()), float(outs["fp32"].abs().max())))
