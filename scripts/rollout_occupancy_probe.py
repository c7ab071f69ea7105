"""C2 rollout: kernel time per ply at K = 20 and 200 plies per launch against the number of workgroups a CU holds at a time
(TRL_ROLLOUT_LDS_PAD: extra LDS per workgroup; the rest of the 4096 games are dispatched as others finish) and pacing.
Question: is the short launch's loss (its last games finish on an emptying chip) smaller when the dispatcher balances the
load, and what does the lower occupancy cost."""
import os
import sys
import statistics

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl

N = 4096
print("pad bytes (wg/CU)  pace   K=20: median us/ply of 9 launches [min..max]      K=200: of 3", flush=True)
for pad, per_cu in ((0, 16), (800, 14), (2200, 12), (5400, 10), (9500, 8)):
    for pace in (1, 0):
        os.environ["TRL_ROLLOUT_LDS_PAD"] = str(pad)
        os.environ["TRL_PACE"] = str(pace)
        e = trl.Engine(N, seed=20260630)
        e.reset()
        e.rollout_random(300)
        out = []
        for k, reps in ((20, 9), (200, 3)):
            ms = []
            for _ in range(reps):
                e.rollout_random(k)
                ms.append(e.last_kernel_ms() * 1e3 / k)
            out.append("%6.2f [%6.2f .. %6.2f]" % (statistics.median(ms), min(ms), max(ms)))
        e.close()
        print("%6d (%2d)        %d     %s      %s" % (pad, per_cu, pace, out[0], out[1]), flush=True)
