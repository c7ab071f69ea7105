"""probe: where k_select / k_expand<2> spend their cycles, summed over the 1024 slots of C3 ticks (needs the
-DTRL_PROFILE build, scripts/_prof/libtrl_prof.so)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
from tetris_rl_amd import engine
L = engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_prof.so"))
engine._lib = L
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
n = 1024
sp = trl.SelfPlay(cfg, net, n, seed=20260630, use_graph=True, record_capacity=65536)
sp.tick(300); sp.synchronize()
buf = (C.c_ulonglong * 16)()
L.trl_debug_mcts_prof(buf, 1)
T = 400
sp.tick(T); sp.synchronize()
L.trl_debug_mcts_prof(buf, 1)
v = np.array(buf[:], dtype=np.float64) / (T * n)
print("k_select, cycles per slot-tick: ctl+new search %.0f | selection walk %.0f | make_move %.0f | legal moves %.0f | features+ctl %.0f | total %.0f"
      % (v[0], v[1], v[2], v[3], v[4], v[:5].sum()))
print("   of which legal-move logits %.0f | normalisation + child initialisation %.0f" % (v[13], v[14]))
print("k_expand<2>: z+dense heads %.0f | policy logits+children %.0f | noise %.0f | backup+fpu %.0f | end of search %.0f | ctl store %.0f | total %.0f"
      % (v[8], v[9], v[10], v[11], v[12], 0, v[8:13].sum()))
