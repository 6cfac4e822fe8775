"""Worker of test_data_parallel_gradients_equal_full_batch: one rank of a data-parallel training step.
Each rank owns a contiguous shard of the chains (l2hmc_amd.dist.shard_bounds) and a replica of the weights;
after GaugeTrainer's single all-reduce every rank holds the full-batch gradient.  Rank 0 saves it."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train import _setup  # noqa: E402
from l2hmc_amd.dist import shard_bounds  # noqa: E402


def toy(out, B, rank, world):
    """Same check for the toy-target trainer (DynamicsTrainer): sharded chains, two all-reduces."""
    from tests.test_gpu_train import _small_setup
    from l2hmc_amd.dynamics_trainer import DynamicsTrainer
    tr, tm, x, z, dx, dz = _small_setup("mog", 50, 5, 0.1, B, "stress")
    lo, hi = shard_bounds(B, world, rank)
    tr = DynamicsTrainer(tr.dynamics, scale=0.1, dist=dist)
    loss, *_ = tr.calc_loss_and_grads(x[lo:hi], z=z[lo:hi], draws_x=tuple(a[lo:hi] for a in dx),
                                      draws_z=tuple(a[lo:hi] for a in dz))
    if rank == 0:
        np.savez(out, grads=tr.grads.cpu().numpy(), loss=float(loss))


def main():
    out, B = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)            # the test box has one GPU: both ranks share it, gloo carries the exchange
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if len(sys.argv) > 3 and sys.argv[3] == "toy":
        toy(out, B, rank, world)
        dist.barrier()
        dist.destroy_process_group()
        return
    tr, tm, x, z, dx, dz = _setup(4, 2, 0.2, B, "mild")
    lo, hi = shard_bounds(B, world, rank)
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    if rank == 1:      # a replica that starts out of sync: the trainer's initial broadcast must repair it
        for net in (tr.dynamics.position_fn, tr.dynamics.momentum_fn):
            net.flat_params()[0].mul_(1.5)
        tr.dynamics.eps = tr.dynamics.eps * 2
    tr = GaugeTrainer(tr.dynamics, dist=dist)           # bucketed, overlapped all-reduce (the default)
    assert tr.bucketed
    shard = dict(z=z[lo:hi], draws_x=tuple(a[lo:hi] for a in dx), draws_z=tuple(a[lo:hi] for a in dz))
    loss, *_ = tr.calc_loss_and_grads(x[lo:hi], 2.5, **shard)
    assert tr.last_bucket_count == 7                    # 3 groups per network + (eps)
    g_bucketed = tr.grads.cpu().numpy().copy()
    tr.bucketed = False                                 # one all-reduce of the whole buffer after the pass
    tr.calc_loss_and_grads(x[lo:hi], 2.5, **shard)
    g_single = tr.grads.cpu().numpy().copy()
    tr.apply_gradients()
    if rank == 0:
        np.savez(out, grads=g_bucketed, grads_single=g_single, loss=float(loss), lr=tr.learning_rate(),
                 w=tr._nets[0].flat_params()[0].cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
