"""probe for a rocprofv3 counter pass: after 300 plies of set-up, 8 x (k_legal, k_step with the first legal move) on 4096
games -- how a ply's vector instructions split between the two legal-placement searches and the move itself"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tetris_rl_amd as trl
e = trl.Engine(4096, seed=20260630)
e.reset()
e.rollout_random(300)
for i in range(8):
    counts, moves = e.legal_moves()
    mv = np.where(counts > 0, moves[:, 0], 0).astype(np.uint16)
    e.step(mv, add_bag=True, check_legal=False)
print("done")
