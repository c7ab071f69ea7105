"""probe: per-slot cycle stamps of the fused search kernel (k_expand_select) for one launch -- a launch lasts as long as its
slowest slot, so the tail matters, not the mean (needs the -DTRL_PROFILE build, scripts/_prof/libtrl_prof.so)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
from tetris_rl_amd import engine
L = engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_prof.so"))
engine._lib = L
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sp = trl.SelfPlay(cfg, trl.AlphaSame(mc), n, seed=20260630, use_graph=True, record_capacity=65536)
assert sp.fused_select
names = {8: "z+heads", 13: "logits", 14: "norm+children", 10: "noise", 11: "backup+fpu", 12: "end", 0: "ctl/new", 1: "select walk",
         2: "make_move", 3: "legal moves", 4: "features"}
order = [8, 13, 14, 10, 11, 12, 0, 1, 2, 3, 4]
acc = []
for rep in range(6):
    sp.tick(37 + rep); sp.synchronize()
    buf = np.zeros((n, 16), dtype=np.uint32)
    L.trl_debug_mcts_slotprof.argtypes = [C.c_void_p, C.c_int]
    assert L.trl_debug_mcts_slotprof(buf.ctypes.data, n) == 0
    acc.append(buf.astype(np.float64))
v = np.concatenate(acc)                                    # [6 launches x n slots][16]
v[:, 9] = 0                                                # (9 = 13 + 14 + root softmax: not added twice)
tot = v[:, order].sum(axis=1)
print("cycles per slot and launch (%d slots, 6 launches): mean %.0f  p50 %.0f  p95 %.0f  p99 %.0f  max %.0f" %
      (n, tot.mean(), np.percentile(tot, 50), np.percentile(tot, 95), np.percentile(tot, 99), tot.max()))
print("%-14s %8s %8s %8s %8s" % ("phase", "mean", "p95", "max", "mean of slowest 2 %"))
slow = np.argsort(-tot)[:max(1, len(tot) // 50)]
for i in order:
    print("%-14s %8.0f %8.0f %8.0f %8.0f" % (names[i], v[:, i].mean(), np.percentile(v[:, i], 95), v[:, i].max(), v[slow, i].mean()))
