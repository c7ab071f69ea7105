"""probe: the 64 / 32-filter fused tower against the PyTorch (MIOpen) tower, ms per 2048 boards; C5-shaped search ticks"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tetris_rl_amd as trl
torch.manual_seed(0)
for blocks, filters in ((20, 64), (10, 32)):
    mc = trl.AlphaSameConfig(blocks=blocks, filters=filters)
    net = trl.AlphaSame(mc).cuda().prepare()
    m = 2048
    boards = (torch.rand((m, 1, 40, 10), device="cuda") < 0.3).float()
    for hip in (True, False):
        net.use_hip_trunk = hip
        net.trunk(boards); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): net.trunk(boards)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
        flop = m * (2 * blocks * 400 * 9 * filters * filters * 2 + 400 * 25 * filters * 2)
        print("b%df%d hip=%s: %.2f ms / %d boards, %.1f TFLOP/s" % (blocks, filters, hip, dt * 1e3, m, flop / dt / 1e12), flush=True)
mc = trl.AlphaSameConfig(blocks=20, filters=64)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model='pytorch', model_config=mc, MAX_ITER=800, training=True, use_forced_playouts_and_policy_target_pruning=True)
for hip in (True, False):
    net.use_hip_trunk = hip
    sp = trl.SelfPlay(cfg, net, 1024, seed=20260630, use_graph=True, record_capacity=65536)
    sp.tick(5); sp.synchronize()
    t0 = time.perf_counter(); sp.tick(40); sp.synchronize(); dt = (time.perf_counter() - t0) / 40
    print("C5 1024 slots hip=%s: %.2f ms/tick -> %.1f moves/s at 800 sims" % (hip, dt * 1e3, 1024 / (800 * dt)), flush=True)
    del sp
# the reference's default net: BaseResNet(8 blocks, 32 filters), evaluation of 1024 positions (2048 boards)
mc = trl.BaseResNetConfig()
net = trl.BaseResNet(mc).cuda().prepare()
n = 1024
grids = (torch.rand((2, n, 400), device="cuda") < 0.3).float()
vec = torch.rand((n, 105), device="cuda")
for hip in (True, False):
    net.use_hip_trunk = hip
    net.head_inputs(grids, vec); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): net.head_inputs(grids, vec)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("BaseResNet default hip=%s: %.2f ms / %d positions" % (hip, dt * 1e3, n), flush=True)
