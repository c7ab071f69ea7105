import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tetris_rl_amd as trl
n = 1024
e = trl.Engine(n, seed=20260630); e.reset(); e.rollout_random(30)
st = e.get_state()
for k in (1, 2, 4, 8, 16, 64):
    ts = []
    for _ in range(4):
        e.set_state(st); e.rollout_random(k); ts.append(e.last_kernel_ms())
    print("rollout plies=%2d: %.1f us total, %.1f us/ply" % (k, min(ts)*1e3, min(ts)*1e3/k), flush=True)
