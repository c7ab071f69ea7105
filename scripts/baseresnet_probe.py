"""probe: search tick with the reference's default nets (BaseResNet / AuxBaseResNet, 8 blocks x 32 filters)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
torch.manual_seed(0)
for name, mc, cls in (("BaseResNet", trl.BaseResNetConfig(), trl.BaseResNet), ("AuxBaseResNet", trl.AuxBaseResNetConfig(), trl.AuxBaseResNet)):
    net = cls(mc)
    cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
    for hip in (True, False) if os.environ.get("TRL_PROBE_TORCH_TOWER") else (True,):
        net.use_hip_trunk = hip
        sp = trl.make_self_play(cfg, net, 1024, seed=20260630, use_graph=True, record_capacity=65536)   # TRL_SLOT_GROUPS=n: n groups
        sp.tick(200); sp.synchronize()
        t0 = time.perf_counter(); sp.tick(200); sp.synchronize(); dt = (time.perf_counter() - t0) / 200
        print("%s 1024 slots, %d groups, fused tower=%s: %.3f ms/tick -> %.0f moves/s at 160 sims" % (name, len(getattr(sp, 'parts', [sp])), hip, dt * 1e3, 1024 / (160 * dt)), flush=True)
        del sp
