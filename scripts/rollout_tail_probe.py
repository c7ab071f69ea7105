"""C2 rollout: when does each slot finish its K plies?  (needs the -DTRL_SLOT_TIMES build:
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
      -DTRL_SLOT_TIMES -o scripts/_prof/libtrl_slot.so tetris-reinforcement-learning_amd/csrc/*.hip)
Prints percentiles of the slots' finishing times (100 MHz chip-wide clock, relative to the first slot's start) for the
launch unpaced and paced: a launch lasts as long as its last slot, the chip is full only until the first slots leave."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tetris_rl_amd as trl
from tetris_rl_amd import engine

L = engine.load_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_prof", "libtrl_slot.so"))
engine._lib = L
N = 4096
buf = (C.c_ulonglong * (2 * N))()
for k in (20, 80, 1000):  # noqa
    for pace, thr, rk in ((0, None, None), (1, None, None), (1, None, "2,8,32"), (1, None, "8,24,48")):
        os.environ["TRL_PACE"] = str(pace)
        if rk:
            os.environ["TRL_PACE_R"] = rk
        else:
            os.environ.pop("TRL_PACE_R", None)
        e = trl.Engine(N, seed=20260630)
        e.reset()
        e.rollout_random(300)
        e.rollout_random(k)
        ms = e.last_kernel_ms()
        L.trl_debug_slot_times(buf, N)
        t = np.array(buf[:], dtype=np.float64).reshape(N, 2)
        t0 = t[:, 0].min()
        start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0          # us
        pct = np.percentile(end, [0, 1, 10, 50, 90, 99, 100])
        print("K %4d pace %d %-22s kernel %8.1f us = %5.1f us/ply | starts: last %.1f us | finish us (min 1%% 10%% 50%% 90%% 99%% max): %s"
              % (k, pace, (thr or "") + " r" + (rk or "-"), ms * 1e3, ms * 1e3 / k, start.max(), " ".join("%8.1f" % x for x in pct)), flush=True)
        print("        mean finish by dispatch age (slot id / 256):", " ".join("%.0f" % end[b * 256:(b + 1) * 256].mean() for b in range(N // 256)),
              "| latest slots:", " ".join("%d" % i for i in np.argsort(end)[-8:]), flush=True)
        e.close()
