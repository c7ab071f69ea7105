"""bench.py's entry for N > 1 (BASELINE.json C4): `python3 bench.py --gpus 2`, started as ONE process the way the driver
starts it, must launch its own ranks, relay rank 0's JSON line and report the world size the process group saw.
Two ranks share this box's one GPU, so the process group is gloo (RCCL refuses two ranks on one device); sizes are cut
so the test takes seconds, the code path is the driver's."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "20", "--warmup", "5", "--games", "512", "--preroll", "60", "--no-cpu-baseline", "--mcts-games", "64",
         "--mcts-ticks", "96", "--mcts-warmup", "16", "--no-c5"]


def _run(gpus, extra_env=None):
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + SMALL, env=env,
                       stdout=subprocess.PIPE, timeout=900)
    return r.returncode, r.stdout.decode()


def test_two_ranks_self_launched():
    rc, out = _run(2, {"TRL_BENCH_BACKEND": "gloo"})
    assert rc == 0, out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, out                                   # ONE JSON line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["dist"] == {"backend": "gloo", "world_size": 2}
    assert line["config"]["env_steps_timed"] > 512 * 20 and line["value"] > 0      # both ranks' steps are in the total
    assert line["scaling"] == "weak" and line["steps"] == 20 and line["warmup"] == 5
    assert line["mcts"]["value"] > 0 and line["mcts"]["config"]["records_to_rank0_mb"] > 0
    assert "cpu_baseline" not in line                             # reported at N = 1 only

    rc1, out1 = _run(1)
    assert rc1 == 0, out1
    one = json.loads(out1.strip().splitlines()[-1])
    assert one["n_gpus"] == 1 and one["dist"]["world_size"] == 1
    # weak scaling: the two-rank job timed twice the env-steps (same games per rank; auto-reset keeps every game live)
    assert abs(line["config"]["env_steps_timed"] - 2 * one["config"]["env_steps_timed"]) <= 0.02 * line["config"]["env_steps_timed"]


def test_rank_failure_is_reported():
    # RCCL refuses two ranks on one device: the ranks fail, the launcher must exit non-zero and print no JSON line
    if __import__("torch").cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    rc, out = _run(2, {"TRL_BENCH_BACKEND": "nccl"})
    assert rc != 0 and out.strip() == ""
