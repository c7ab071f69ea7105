"""timing of the 16-filter tower for experimental builds: python scripts/tower16_exp.py lib1.so lib2.so ... (each in a
fresh process: a process can hold one build of the library)"""
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
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16)).cuda().prepare()
for m in (2048, 1024):
    boards = (torch.rand((m, 1, 40, 10), device="cuda") < 0.3).float()
    for _ in range(5): net.trunk(boards)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): net.trunk(boards)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("%-40s %5d boards: %.3f ms  %.1f TFLOP/s" % (os.path.basename(lib), m, dt * 1e3, m * 36.9e6 / dt / 1e12), flush=True)
