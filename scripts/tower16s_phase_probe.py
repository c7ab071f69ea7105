"""probe: cycles of workgroups 0 and 300 of the split 16-filter tower (k_trunk16s) per phase and wave, for a launch of
n boards (1 board per CU at 256, 2 at 512); needs the -DTRL_PROFILE build (scripts/README.md)"""
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
assert net.tower_mode == "split"
names = ["H1 split+write", "barrier", "conv a + B loads", "barrier", "reduce + split+write", "barrier", "conv b + x", "barrier"]
for n in (256, 512, 1024):
    boards = (torch.rand((n, 1, 40, 10), device="cuda") < 0.3).float()
    for _ in range(3):
        net.trunk(boards)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 160)()
    L.trl_debug_tower_prof(buf)
    v = np.array(buf[:], dtype=np.float64).reshape(2, 8, 10)
    print("== %d boards" % n)
    for k, wg in enumerate((0, 300)):
        if wg >= n:
            continue
        print("workgroup %d: per block (cycles, %d blocks), waves 0..3:" % (wg, blocks))
        for i, nm in enumerate(names):
            print("  %-22s" % nm, " ".join("%8.0f" % (v[k, w, i] / blocks) for w in range(4)))
        print("  %-22s" % "block loop", " ".join("%8.0f" % (v[k, w, 8] / blocks) for w in range(4)))
        print("  prologue (zero fill, board load, conv1): %.0f cycles; block loop total %.0f" % (v[k, 0, 9], v[k, 0, 8]))
