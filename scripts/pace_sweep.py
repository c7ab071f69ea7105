import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl
N = 4096
for ranks in ("4,16,40", "2,8,32", "8,24,48", "1,4,16", "6,20,44", "4,12,28", "3,10,24", "off"):
    if ranks == "off":
        os.environ["TRL_PACE"] = "0"
    else:
        os.environ["TRL_PACE"] = "1"; os.environ["TRL_PACE_R"] = ranks
    e = trl.Engine(N, seed=20260630); e.reset(); e.rollout_random(300)
    res = []
    for k, reps in ((20, 15), (1000, 2)):
        ms = []
        for _ in range(reps):
            e.rollout_random(k); ms.append(e.last_kernel_ms() * 1e3 / k)
        res.append("K=%d %.2f us/ply" % (k, statistics.median(ms)))
    e.close()
    print(ranks, "  ".join(res), flush=True)
