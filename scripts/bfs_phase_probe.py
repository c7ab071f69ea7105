"""probe: where a legal-move search spends its cycles (needs the -DTRL_PROFILE build, scripts/_prof/libtrl_prof.so)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tetris_rl_amd as trl
from tetris_rl_amd import engine
L = engine.load_library(os.environ.get("TRL_ALT_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_prof.so"))
engine._lib = L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024         # 256: one workgroup per CU, a search alone on its SIMD (latency)
e = trl.Engine(n, seed=20260630); e.reset(); e.rollout_random(30)
buf = (C.c_ulonglong * 16)()
L.trl_debug_prof(buf, 1)
for _ in range(5):
    e.legal_moves()
L.trl_debug_prof(buf, 1)
v = np.array(buf[:], dtype=np.float64)
names = ["valid build", "classify", "flood rows", "compact", "kick passes", "convert", "total"]
nb = v[11]
print("searches", int(nb), " per search: batches %.1f  flood rows %.1f  kick passes %.1f" % (v[8]/nb, v[9]/nb, v[10]/nb))
for i, nm in enumerate(names):
    print("  %-12s %8.0f cycles/search  (%.1f%%)" % (nm, v[i]/nb, 100*v[i]/v[6]))

# same breakdown inside the persistent rollout kernel (warm caches, state resident in LDS)
e.rollout_random(5)
L.trl_debug_prof(buf, 1)
e.rollout_random(50)
L.trl_debug_prof(buf, 1)
v = np.array(buf[:], dtype=np.float64)
nb = v[11]
print("rollout: searches", int(nb), " per search: batches %.1f  flood rows %.1f  kick passes %.1f" % (v[8]/nb, v[9]/nb, v[10]/nb))
for i, nm in enumerate(names):
    print("  %-12s %8.0f cycles/search  (%.1f%%)" % (nm, v[i]/nb, 100*v[i]/v[6]))
