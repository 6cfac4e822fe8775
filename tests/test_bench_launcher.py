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


def test_chain_partition_of_every_config():
    """contiguous blocks of B / world_size chains; configs 4 and 5 are 8192 and 16384 chains over the
    node (1024 and 2048 per GPU at 8 GPUs), config 3 weak-scales 2048 per GPU or strong-scales its 2048."""
    for cfg, glob, per in ((4, 8192, 1024), (5, 16384, 2048)):
        blocks = [bench.rank_chains(cfg, 8, r, "strong") for r in range(8)]
        assert [b[:2] for b in blocks] == [(r * per, (r + 1) * per) for r in range(8)]
        assert all(b[2] == glob for b in blocks)
        assert bench.rank_chains(cfg, 1, 0, "strong") == (0, glob, glob)           # one GPU: the whole batch
        assert bench.rank_chains(cfg, 1, 0, "weak") == (0, per, per)               # one GPU: one shard of the 8
    assert [bench.rank_chains(3, 4, r, "weak")[:2] for r in range(4)] == [(2048 * r, 2048 * (r + 1)) for r in range(4)]
    assert bench.rank_chains(3, 4, 0, "weak")[2] == 8192
    assert [bench.rank_chains(3, 4, r, "strong") for r in range(4)] == [(512 * r, 512 * (r + 1), 2048) for r in range(4)]
    # a batch that does not divide: remainder on the first ranks, nothing lost, nothing doubled
    cuts = [bench.rank_chains(3, 3, r, "strong")[:2] for r in range(3)]
    assert cuts == [(0, 683), (683, 1366), (1366, 2048)]
    assert bench.CONFIGS[4]["scaling"] == bench.CONFIGS[5]["scaling"] == "strong" and bench.CONFIGS[3]["scaling"] == "weak"


def test_algorithmic_flops_match_the_survey_sizes_table():
    """SURVEY.md 8, sizes table: FLOPs per chain-LF step = 8 x MACs per net call."""
    assert 8 * bench.config_macs(1)[0] == 1760 and 8 * bench.config_macs(2)[0] == 24800
    assert 8 * bench.config_macs(3)[0] == 4726784
    assert 8 * bench.config_macs(4)[0] == 35930112
    assert 8 * bench.config_macs(5)[0] == 1208090624


def test_executed_flops_never_exceed_the_reference_count():
    """Every `frac` of the bench line prices EXECUTED FLOPs (a fraction above 1 was printed for
    cfg 5 because the reference's count was divided by the time of kernels that skip part of it).  The executed count
    is at most the reference graph's for every config and every kernel path, equals it for the toy kernels, and
    reproduces the counter-calibrated MFMA count of the whole-step kernel (87,031,808 of 94,371,840 instructions
    per launch at cfg 3: profiles/r02_pmc_fused_kernel.json)."""
    for cfg in (1, 2, 3, 4, 5):
        ref = 8 * bench.config_macs(cfg)[0]
        for fused in (False, True):
            for active in (False, True):
                ex = 2.0 * bench.executed_macs_per_lf(cfg, fused, active)
                assert 0 < ex <= ref, (cfg, fused, active, ex, ref)
                if cfg in (1, 2):
                    assert ex == ref
    # cfg 3, whole-step kernel: MFMA work only (the 2 H time-term products per call run on the VALU)
    H, D, N = 512, 128, 10
    mfma_ref = 4 * (2 * D * H + H * H + 3 * H * D)
    mfma_ex = bench.executed_macs_per_lf(3, True, False) - 4 * 2 * H
    assert abs(mfma_ex / mfma_ref - 87031808 / 94371840) < 1e-6
    # the library's choice of the active-column form, mirrored in bench.heads_use_active_columns
    assert bench.heads_use_active_columns(2 * 1024, 512, 1024, 1024)          # cfg 4 shard: 64 x 32 tiles, split % 64 == 0
    assert bench.heads_use_active_columns(2 * 2048, 2048, 8192, 2048)         # cfg 5 shard: 128 x 64 tiles, split % 128 == 0
    assert not bench.heads_use_active_columns(2 * 37, 128, 512, 37)           # split inside a tile
    assert bench.heads_use_active_columns(2 * 64, 128, 512, 64) and not bench.heads_use_active_columns(2 * 96, 128, 512, 96)
    assert not bench.heads_use_active_columns(2 * 16448, 2048, 8192, 16448)   # 128-row tiles, split % 128 == 64


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


def test_config4_partition_over_two_ranks():
    """`--config 4 --gpus 2`: the 8192 chains of BASELINE.json configs[3] in two contiguous shards, through the
    real launcher; config 3 with `--scaling strong` cuts its 2048."""
    for args, want, glob in ((["--config", "4"], [[0, 4096], [4096, 8192]], 8192),
                             (["--config", "3", "--scaling", "strong"], [[0, 1024], [1024, 2048]], 2048)):
        r = _run(["--gpus", "2", "--rendezvous-only", *args])
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        assert out["chains"] == want and out["global_batch"] == glob and out["ranks_seen"] == 2, out


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
