"""probe: cycles of workgroup 0 of the 64-filter tower per phase and wave (needs the -DTRL_PROFILE build,
scripts/_prof/libtrl_prof.so; see scripts/README.md)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
from tetris_rl_amd import engine
L = engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_prof.so"))
engine._lib = L
torch.manual_seed(0)
blocks = 20
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=64)).cuda().prepare()
boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
for _ in range(3):
    net.trunk(boards)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 160)()
L.trl_debug_tower_prof(buf)
v = np.array(buf[:], dtype=np.float64).reshape(2, 8, 10)
names = ["write planes", "barrier 1", "conv", "x update", "barrier 2", "layer loop total"]
for k, wg in enumerate((0, 1500)):
    print("workgroup %d: per conv (cycles, %d convs), waves 0..7:" % (wg, 2 * blocks))
    for i, nm in enumerate(names):
        print("  %-18s" % nm, " ".join("%8.0f" % (v[k, w, i] / (2 * blocks)) for w in range(8)))
