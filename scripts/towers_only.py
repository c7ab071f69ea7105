"""probe for rocprofv3 counter passes: four launches each of k_trunk16s (AlphaSame(10,16), 1024 boards) and
k_trunk_wide_s<64> (AlphaSame(20,64), 1024 boards), nothing else"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
boards = (torch.rand((1024, 1, 40, 10), device="cuda") < 0.3).float()
for blocks, filters in ((10, 16), (20, 64)):
    net = trl.AlphaSame(trl.AlphaSameConfig(blocks=blocks, filters=filters)).cuda().prepare()
    for _ in range(4):
        net._trunk_hip(boards, net._folded["packed_tower"])
    torch.cuda.synchronize()
print("done")
