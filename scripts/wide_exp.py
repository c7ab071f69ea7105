"""timing experiments on the split towers: python scripts/wide_exp.py [library built with -DTRL_EXP=n]
k_trunk_wide_s: TRL_EXP = 1 weights not re-read per K-step; 2 fragments not re-read; 3 only one tile's split + write;
4 no per-block parameter loads.  k_trunk16s: 5 no barriers; 6 fragments not re-read; 7 only one tile's split + write.
Results are wrong; the time shows what that ingredient costs -- with care: a variant that leaves the planes zero also lets
the chip clock higher (3 and 7 overstate the cost of the write phase for that reason).  Build:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
-DTRL_EXP=n -o scripts/_prof/libexpn.so tetris-reinforcement-learning_amd/csrc/*.hip"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import tetris_rl_amd as trl
from tetris_rl_amd import engine
if len(sys.argv) > 1:
    engine._lib = engine.load_library(sys.argv[1])
torch.manual_seed(0)
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=20, filters=64)).cuda().prepare()
net16 = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16)).cuda().prepare()
boards = (torch.rand((1024, 1, 40, 10), device="cuda") < 0.3).float()
for name, m in (("b20f64", net), ("b10f16", net16)):
    line = []
    for n in (256, 1024):
        b = boards[:n].contiguous()
        m.trunk(b); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m.trunk(b)
        e1.record(); torch.cuda.synchronize()
        line.append("%d: %.3f ms" % (n, e0.elapsed_time(e1) / 10))
    print(sys.argv[1] if len(sys.argv) > 1 else "default", name, "  ".join(line), flush=True)
