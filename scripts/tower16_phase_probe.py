"""probe: cycles of workgroup 0 of the 16-filter tower per phase and wave (needs the -DTRL_PROFILE build)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
from tetris_rl_amd import engine
L = engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_prof.so"))
engine._lib = L
torch.manual_seed(0)
blocks = 10
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=16)).cuda().prepare()
boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
for _ in range(3):
    net.trunk(boards)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 160)()
L.trl_debug_tower_prof(buf)
v = np.array(buf[:], dtype=np.float64).reshape(2, 8, 10)
names = ["H1 write", "barrier", "conv a", "barrier", "reduce+write", "barrier", "conv b + x", "barrier"]
for k, wg in enumerate((0, 1500)):
    print("workgroup %d: per block (cycles, %d blocks), waves 0..3:" % (wg, blocks))
    for i, nm in enumerate(names):
        print("  %-14s" % nm, " ".join("%8.0f" % (v[k, w, i] / blocks) for w in range(4)))
    print("  %-14s" % "block loop", " ".join("%8.0f" % (v[k, w, 8] / blocks) for w in range(4)))
    print("  prologue (zero fill, board load, conv1): %.0f cycles; block loop total %.0f" % (v[k, 0, 9], v[k, 0, 8]))
