"""probe for rocprofv3 counter passes: one C2-shaped rollout launch (4096 games x 200 plies), nothing else"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl
e = trl.Engine(4096, seed=20260630); e.reset(); e.rollout_random(100)
e.rollout_random(200)
print("kernel ms", e.last_kernel_ms())
