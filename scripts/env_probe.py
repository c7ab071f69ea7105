"""probe: kernel time of legal-move search and make_move alone at several batch sizes"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tetris_rl_amd as trl
for n in (256, 1024, 4096, 16384):
    e = trl.Engine(n, seed=20260630)
    e.reset()
    e.rollout_random(30)
    c, m = e.legal_moves()
    ts = []
    for _ in range(5):
        e.legal_moves(); ts.append(e.last_kernel_ms())
    mv = m[np.arange(n), 0].copy()
    st = e.get_state()
    t2 = []
    for _ in range(5):
        e.set_state(st); e.step(mv, check_legal=False); t2.append(e.last_kernel_ms())
    t3 = []
    for _ in range(3):
        e.set_state(st); e.rollout_random(50); t3.append(e.last_kernel_ms() / 50)
    print("n=%5d legal %.1f us  step(no check) %.1f us  rollout/ply %.1f us  mean legal %.1f" % (n, min(ts)*1e3, min(t2)*1e3, min(t3)*1e3, c.mean()), flush=True)
    e.close()
