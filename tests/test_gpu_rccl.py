"""RCCL on the one GPU of the test box: a one-rank `nccl` group still runs every collective of the sharded flow
(l2hmc_amd/dist.py:active with L2HMC_COLLECTIVES_AT_WORLD1=1), so the calls bench.py and the trainers make at
N > 1 -- group creation with a device id, the per-step 12-byte all-reduce on the side stream, the bucketed
gradient all-reduces issued from the backward pass's host callback, barriers, the MAX reduction of the timing --
are exercised through the real backend.  (Multi-rank values are covered by the gloo tests: tests/test_dist_gloo.py,
tests/test_gpu_train.py::test_data_parallel_gradients_equal_full_batch.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *args):
    env = dict(os.environ, **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline", "--no-trained-ess", "--no-roofline", *args],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_flow_through_a_one_rank_rccl_group_matches_the_plain_run():
    plain = _run({})
    rccl = _run({"L2HMC_COLLECTIVES_AT_WORLD1": "1", "MASTER_PORT": "29541"})
    for out in (plain, rccl):
        assert out["n_gpus"] == 1 and out["value"] > 0
        assert "error" not in out["config"]["train_step"], out["config"]["train_step"]
        assert "error" not in out["config"]["mog_cfg2"], out["config"]["mog_cfg2"]
    # same seeds, and a one-rank sum changes nothing: the chains and the training loss are the same numbers
    assert rccl["config"]["mean_accept_prob"] == plain["config"]["mean_accept_prob"]
    assert rccl["config"]["train_step"]["loss"] == plain["config"]["train_step"]["loss"]
    assert rccl["config"]["train_step"]["grad_buckets"] == 7          # the overlapped, bucketed exchange ran
    assert plain["config"]["train_step"]["grad_buckets"] == 0
