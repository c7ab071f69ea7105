"""C3 as bench.py ticks it (1024 slots, AlphaSame(10,16), slot groups as selfplay.slot_groups decides), ticks only: the
program rocprofv3 traces for scripts/tick_overlap.py.   usage: python scripts/mcts_c3_ticks.py [ticks] [warmup]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tetris_rl_amd as trl
if os.environ.get("TRL_ALT_LIB"):                   # A/B runs: another build of the library
    from tetris_rl_amd import engine
    engine._lib = engine.load_library(os.environ["TRL_ALT_LIB"])
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 800
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 400
torch.manual_seed(0)
mc = trl.AlphaSameConfig(blocks=10, filters=16)
net = trl.AlphaSame(mc)
cfg = trl.Config(visual=False, model="pytorch", model_config=mc, MAX_ITER=160, training=True)
sp = trl.make_self_play(cfg, net, 1024, seed=20260630, auto_reset=True, use_graph=True, record_capacity=1 << 17)
sp.tick(warm)
sp.synchronize()
t0 = time.perf_counter()
sp.tick(ticks)
sp.synchronize()
dt = time.perf_counter() - t0
print("%d groups, %d ticks: %.4f ms per round of ticks" % (len(getattr(sp, "parts", [sp])), ticks, dt / ticks * 1e3), flush=True)
