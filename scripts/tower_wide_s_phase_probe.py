"""probe: cycles of workgroup 0 of the split wide tower (k_trunk_wide_s<64>, 20 blocks) per phase and wave for a launch of
n boards; needs the -DTRL_PROFILE build (scripts/README.md).  The counter is s_memtime: compare phases within one launch."""
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
assert net.tower_mode == "split"
names = ["split + write", "barrier", "conv", "corr + residual", "barrier"]
for n in (256, 1024):
    boards = (torch.rand((n, 1, 40, 10), device="cuda") < 0.3).float()
    for _ in range(2):
        net.trunk(boards)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); net.trunk(boards); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * 160)()
    L.trl_debug_tower_prof(buf)
    v = np.array(buf[:], dtype=np.float64).reshape(2, 8, 10)
    print("== %d boards, %.3f ms; workgroup 0, per layer (cycles, %d layers), waves 0..7:" % (n, e0.elapsed_time(e1), 2 * blocks))
    for i, nm in enumerate(names):
        print("  %-18s" % nm, " ".join("%7.0f" % (v[0, w, i] / (2 * blocks)) for w in range(8)))
    print("  %-18s" % "layer", " ".join("%7.0f" % (v[0, w, 8] / (2 * blocks)) for w in range(8)))
    print("  matrix instructions per SIMD and layer: 2 waves x 12.5 tiles x 54 x 16 cycles = 21600")
