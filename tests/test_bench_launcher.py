"""bench.py's own multi-rank launcher (`python bench.py --gpus N` as ONE process): a rank that never joins the others must
not hang the run -- after --rank-timeout the parent names the ranks that were still alive, ends every process it started
and exits non-zero without a JSON line.  Runs without a GPU: the stalled ranks stall before they touch one."""
import os
import subprocess
import sys
import time
import uuid

import psutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _left_over(tag):
    out = []
    for p in psutil.process_iter(["pid"]):
        try:
            if p.environ().get("TRL_BENCH_TEST_TAG") == tag:
                out.append(p.pid)
        except (psutil.Error, OSError):
            pass
    return out


def _run(stall, timeout_s, extra=()):
    tag = uuid.uuid4().hex
    env = dict(os.environ, TRL_BENCH_TEST_STALL_RANK=stall, TRL_BENCH_TEST_TAG=tag, TRL_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rank-timeout", str(timeout_s),
                        "--no-cpu-baseline", "--no-mcts", "--steps", "5", "--warmup", "1", "--games", "64", "--preroll", "0"]
                       + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    return r.returncode, r.stdout.decode(), r.stderr.decode(), time.monotonic() - t0, tag


def test_ranks_that_never_join_are_ended_at_the_timeout():
    rc, out, err, dt, tag = _run("0,1", 4)
    assert rc == 124 and out.strip() == "", (rc, out, err[-2000:])
    assert "ranks [0, 1] still running after --rank-timeout" in err
    assert dt < 60
    time.sleep(0.5)
    assert _left_over(tag) == []                                # procs_at_end: nothing of this run is left


def test_one_stalled_rank_does_not_outlive_a_failed_one():
    # without a GPU rank 0 fails at once ("needs an MI355X"), on a GPU box it waits for rank 1 in the rendezvous until
    # the timeout: either way the launcher exits non-zero, prints no line and leaves no process behind
    rc, out, err, dt, tag = _run("1", 20)
    assert rc != 0 and out.strip() == "", (rc, out, err[-2000:])
    assert dt < 120
    time.sleep(0.5)
    assert _left_over(tag) == []
