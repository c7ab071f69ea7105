"""timing of the C2 rollout for several builds of the library: python scripts/rollout_exp.py lib1.so ... (one process each)"""
import sys, os, subprocess, time
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].startswith("--one=")):
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one=" + lib])
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl
from tetris_rl_amd import engine
lib = sys.argv[1][6:]
engine._lib = engine.load_library(os.path.abspath(lib))
e = trl.Engine(4096, seed=20260630); e.reset(); e.rollout_random(100)
steps = e.rollout_random(1000)
ms = e.last_kernel_ms()
print("%-28s %.1f M env-steps/s  (%.2f us per ply)" % (os.path.basename(lib), steps / ms / 1e3, ms * 1e3 / 1000), flush=True)
