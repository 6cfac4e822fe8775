"""RCCL on the one GPU of the test box: a one-rank `nccl` group still runs every collective of the sharded flow
(l2hmc_amd/dist.py:active with L2HMC_COLLECTIVES_AT_WORLD1=1), so the calls bench.py and the trainers make at
N > 1 -- group creation with a device id, the per-step 12-byte all-reduce on the side stream, the initial weight
broadcast, the bucketed gradient all-reduces issued from the backward pass's host callback, barriers -- are
exercised through the real backend, in this process.  A one-rank sum changes no value, so the sampler and the
trainer must reproduce the group-less run bit for bit.  (Multi-rank values are covered by the gloo tests:
tests/test_dist_gloo.py, tests/test_gpu_train.py::test_data_parallel_gradients_equal_full_batch,
tests/test_bench_launcher.py.)"""
import os

import numpy as np
import pytest
import torch

from tests import helpers as H


def _run(dist):
    from l2hmc_amd import GaugeSampler
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    T = X = 8
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, 3, 0.1, xp, vp)
    dyn = H.gauge_hip(T, X, 3, 0.1, xp, vp, orc.mask, 64)
    x = torch.as_tensor(np.random.default_rng(7).uniform(0, 2 * np.pi, (64, 2 * T * X)), dtype=torch.float32,
                        device="cuda")
    smp = GaugeSampler(dyn, dist=dist)
    xs = x
    for _ in range(5):
        xs = smp.step(xs, 2.0)[0]
    acc = smp.stats.mean_accept()
    tr = GaugeTrainer(dyn, lr_init=1e-4, dist=dist)
    losses = [float(tr.train_step(x, 2.0)[0]) for _ in range(3)]
    torch.cuda.synchronize()
    return xs.cpu(), acc, float(smp.stats.total[2]), losses, tr.grads.cpu().clone(), int(tr.last_bucket_count), tr


@pytest.mark.gpu
def test_sampler_and_trainer_through_a_one_rank_rccl_group_match_the_plain_run(monkeypatch):
    import torch.distributed as dist
    plain = _run(None)
    assert plain[5] == 0                                               # no group: one local gradient buffer
    monkeypatch.setenv("L2HMC_COLLECTIVES_AT_WORLD1", "1")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        dist.barrier()
        rccl = _run(dist)
        assert rccl[6].dist is not None and rccl[6]._side is not None  # the collectives were really issued ...
        assert rccl[5] == 7                                            # ... the gradients in seven overlapped buckets
        t = torch.tensor([1.5], device="cuda", dtype=torch.float64)    # bench.py's MAX reduction of the timing
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
        dist.barrier()
    finally:
        dist.destroy_process_group()
    assert torch.equal(plain[0], rccl[0])                              # same chains
    assert plain[1] == rccl[1] and plain[2] == rccl[2] == 5 * 64       # same accept statistics, every chain counted
    assert plain[3] == rccl[3]                                         # same training losses
    assert torch.equal(plain[4], rccl[4])                              # same gradients, bit for bit
