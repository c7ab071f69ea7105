"""probe: where the closure form of the legal-placement search spends its cycles, standalone (k_legal) and inside the
rollout kernel (needs the -DTRL_PROFILE build, scripts/_prof/libtrl_prof.so)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tetris_rl_amd as trl
from tetris_rl_amd import engine
L = engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_prof.so"))
engine._lib = L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
e = trl.Engine(n, seed=20260630); e.reset(); e.rollout_random(300)
buf = (C.c_ulonglong * 16)()
names = ["set-up (free columns, validity, edges)", "fixed point", "kick rounds", "planes + move list"]
def report(tag):
    L.trl_debug_cprof(buf, 1)
    v = np.array(buf[:], dtype=np.float64)
    nb = v[8]
    tot = v[:4].sum()
    print("%s: %d searches; per search %.1f fixed-point steps, %.2f kick rounds, %.1f kick probes (direction x kick)" %
          (tag, nb, v[9] / nb, v[10] / nb, v[11] / nb))
    print("  kick rounds: %.1f source cells on average; %.1f %% of the rounds have <= 2 cells, %.1f %% 3..6, %.1f %% 7..12; rounds after "
          "the second: %.2f per search, %.0f %% of them with <= 6 cells" %
          (v[12] / v[10], 100 * v[13] / v[10], 100 * v[14] / v[10], 100 * v[15] / v[10], v[5] / nb, 100 * v[6] / max(v[5], 1)))
    for i, nm in enumerate(names):
        print("  %-40s %8.0f cycles/search  (%.1f %%)" % (nm, v[i] / nb, 100 * v[i] / tot))
L.trl_debug_cprof(buf, 1)
for _ in range(4):
    e.legal_moves()
report("k_legal")
e.rollout_random(100)
report("k_rollout")
