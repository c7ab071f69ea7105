"""C5 as bench.py ticks it (1024 slots, AlphaSame(20,64), MAX_ITER=800, forced playouts + pruning), ticks only, in
TRL_SLOT_GROUPS groups.   usage: TRL_SLOT_GROUPS=2 python scripts/mcts_c5_ticks.py [ticks] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 600
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=20, filters=64)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=800, training=True,
                 use_forced_playouts_and_policy_target_pruning=True)
sp = trl.make_self_play(cfg, net, 1024, seed=20260630, auto_reset=True, use_graph=True, record_capacity=1 << 16)
sp.tick(warm)
sp.synchronize()
st0 = sp.stats()
t0 = time.perf_counter()
sp.tick(ticks)
sp.synchronize()
dt = time.perf_counter() - t0
st1 = sp.stats()
sims = st1["simulations"] - st0["simulations"]
print("%d groups, %d ticks: %.4f ms per round of ticks, %.0f simulations/s, %.1f %% of the leaf boards reused" %
      (len(getattr(sp, "parts", [sp])), ticks, dt / ticks * 1e3, sims / dt,
       100.0 * (st1["boards_reused"] - st0["boards_reused"]) / max(sims, 1)), flush=True)
