"""The N>1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.
Covers what bench.py does across ranks: contiguous chain shards, replicated
state broadcast from rank 0, one fused all-reduce of the per-step scalar sums."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from l2hmc_amd.dist import StepStats, broadcast_state, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        B = 10
        lo, hi = shard_bounds(B, world, rank)
        rng = np.random.default_rng(5)                 # same stream on every rank = the global batch
        p_all = torch.tensor(rng.uniform(size=(4, B)), dtype=torch.float32)
        dq_all = torch.tensor(rng.integers(0, 3, size=(4, B)), dtype=torch.float32)
        w = torch.full((7,), float(rank + 1))           # "weights": rank 0's copy must win
        broadcast_state([w], dist, src=0)
        # one all-reduce per step, three steps per all-reduce (4 steps: one full block + a flushed rest), the default
        got = []
        for every in (1, 3, None):
            stats = StepStats("cpu", dist) if every is None else StepStats("cpu", dist, reduce_every=every)
            for step in range(4):
                stats.push(p_all[step, lo:hi], dq_all[step, lo:hi])
            assert len(stats._unreduced) == {1: 0, 3: 1, None: 4}[every]      # steps still waiting for their collective
            got.append((stats.mean_accept(), stats.mean_abs_dq(), float(stats.total[2])))
            assert not stats._unreduced and not stats._pending
        assert got[0] == got[1] == got[2]               # the same global sums however the steps are grouped
        q.put((rank, *got[0], w.tolist(), float(p_all.mean()), float(dq_all.mean())))
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_stats_match_global_means():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, acc, dq, n, w, want_acc, want_dq in res:
        assert n == 4 * 10                          # every chain counted once per step, on every rank
        assert abs(acc - want_acc) < 1e-6 and abs(dq - want_dq) < 1e-6
        assert w == [1.0] * 7                       # replicated from rank 0


def test_single_process_stats_need_no_collective():
    s = StepStats("cpu", None)
    s.push(torch.tensor([0.5, 1.0]), torch.tensor([0., 2.]))
    assert s.mean_accept() == 0.75 and s.mean_abs_dq() == 1.0
