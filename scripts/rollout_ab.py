"""C2 rollout timing for A/B builds: 4096 games, 300 plies of set-up, then 9 launches of 20 plies and 3 of 1000 (medians).
usage: python scripts/rollout_ab.py [path of another libtrl_hip.so]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl
from tetris_rl_amd import engine
if len(sys.argv) > 1:
    engine._lib = engine.load_library(sys.argv[1])
e = trl.Engine(4096, seed=20260630)
e.reset()
e.rollout_random(300)
out = []
for k, reps in ((20, 9), (1000, 3)):
    ms = []
    for _ in range(reps):
        e.rollout_random(k)
        ms.append(e.last_kernel_ms() * 1e3 / k)
    out.append("K=%d: %.2f us/ply (%.1f M steps/s)" % (k, statistics.median(ms), 4096 / statistics.median(ms)))
print((sys.argv[1] if len(sys.argv) > 1 else "default build") + "  " + "   ".join(out), flush=True)
