"""C2 rollout: kernel time per ply against launch length K, game phase (plies played before the launch) and pacing
(TRL_PACE: 0 = off, 1 = on; TRL_PACE_R: ranks of 64 that get issue priority 3 / 2 / 1).  Answers where the short-launch
penalty of the driver's `--steps 20 --warmup 5` comes from: the phase of the games (all start together, random play
ends a game after ~50 plies) does not matter; the launch length does (profiles/r02/)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tetris_rl_amd as trl

N = 4096


def run(preroll, k, pace, reps=5, ranks=None):
    os.environ["TRL_PACE"] = str(pace)
    if ranks:
        os.environ["TRL_PACE_R"] = ranks
    else:
        os.environ.pop("TRL_PACE_R", None)
    e = trl.Engine(N, seed=20260630)
    e.reset()
    if preroll:
        e.rollout_random(preroll)
    ms = []
    for _ in range(reps):
        e.rollout_random(k)
        ms.append(e.last_kernel_ms())
    e.close()
    return ms


print("preroll  K    pace(ranks)  us/ply per launch (launches of K plies back to back)", flush=True)
for preroll in (5, 300):
    for k in (5, 20, 80, 320, 1000):
        for pace, ranks in ((0, None), (1, None), (1, "2,8,32")):
            ms = run(preroll, k, pace, reps=3 if k == 1000 else 5, ranks=ranks)
            print("%5d  %5d  %d %-10s  %s" % (preroll, k, pace, ranks or ("4,16,40" if pace else "-"),
                                             "  ".join("%6.1f" % (m * 1e3 / k) for m in ms)), flush=True)
