"""timing of the 64-filter tower for several builds of the library (one process each)"""
import sys, os, subprocess, time
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].startswith("--one=")):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one=" + lib])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
from tetris_rl_amd import engine
lib = sys.argv[1][6:]
engine._lib = engine.load_library(os.path.abspath(lib))
torch.manual_seed(0)
for blocks, filters in ((20, 64), (10, 32)):
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=filters)).cuda().prepare()
    boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
    for _ in range(2): net.trunk(boards)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): net.trunk(boards)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    flop = 2048 * (2 * blocks * 400 * 9 * filters * filters * 2 + 400 * 25 * filters * 2)
    print("%-24s b%df%d: %.2f ms, %.1f TFLOP/s" % (os.path.basename(lib), blocks, filters, dt * 1e3, flop / dt / 1e12), flush=True)
