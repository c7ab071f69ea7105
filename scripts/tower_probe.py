"""probe for rocprofv3 counter passes: a few launches of the fused towers (16 and 64 filters), nothing else"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
for blocks, filters in ((20, 64), (10, 16)):
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=filters)).cuda().prepare()
    boards = (torch.rand((2048, 1, 40, 10), device="cuda") < 0.3).float()
    for _ in range(4):
        net.trunk(boards)
    torch.cuda.synchronize()
