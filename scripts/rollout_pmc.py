"""probe for rocprofv3 counter passes: C2-shaped rollout launches of 20, 20 and 1000 plies (4096 games, after 300 plies of
set-up), nothing else.  k_rollout dispatches in order: 300-ply set-up, 5-ply warm-up, 20, 20, 1000."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl
e = trl.Engine(4096, seed=20260630)
e.reset()
e.rollout_random(300)
e.rollout_random(5)
for k in (20, 20, 1000):
    e.rollout_random(k)
    print("plies", k, "kernel ms", e.last_kernel_ms())
