"""bench.py must start from a bare `python bench.py --gpus N`: the parent decides, before any GPU call,
whether it is a rank (WORLD_SIZE set by torch.distributed.run) or has to start the ranks itself, as child
processes.  The 2-rank run below goes through the real launcher on CPU (gloo, 127.0.0.1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_decision():
    assert bench.launch_command(1, {}, ["--gpus", "1"]) is None                 # N = 1: run in place
    assert bench.launch_command(4, {"WORLD_SIZE": "4"}, ["--gpus", "4"]) is None   # already a rank
    cmd = bench.launch_command(4, {}, ["--gpus", "4", "--steps", "7"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7"]
    free = bench.launch_command(2, {}, [])                                       # picks a free port by itself
    assert 1024 < int(free[free.index("--master-port") + 1]) < 65536


def _run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bare_invocation_starts_two_ranks_over_gloo():
    r = _run(["--gpus", "2", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["max_elapsed_s"] >= 0.02


def test_failing_rank_gives_nonzero_exit():
    # no GPU in this container: each rank stops with "needs a GPU", and the parent must report the failure
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("needs a box without a GPU")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0
    assert "needs a GPU" in (r.stderr + r.stdout)


def test_rank_count_mismatch_is_reported():
    r = _run(["--gpus", "2", "--rendezvous-only"], extra_env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
