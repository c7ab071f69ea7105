"""probe for rocprofv3 counter passes: a few launches of the 16-filter fused tower, nothing else"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
net = trl.AlphaSame(trl.AlphaSameConfig(blocks=10, filters=16)).cuda().prepare()
boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
for _ in range(4):
    net.trunk(boards)
torch.cuda.synchronize()
