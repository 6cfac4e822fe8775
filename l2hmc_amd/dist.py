"""Multi-GPU plumbing for the sampler: chains are independent, so each rank
owns a contiguous block of chains and a replica of the weights / masks; the
only exchange is one fused all-reduce of per-step scalar sums (the reduce_mean
of gauge_model.py:795 across shards).  `torch.distributed` with backend "nccl"
is RCCL over xGMI on ROCm; the same code runs on "gloo" for CPU tests."""
import os

import torch


def active(dist):
    """`dist` if its collectives are to be issued, else None.  A single rank has nothing to exchange and skips
    them -- unless L2HMC_COLLECTIVES_AT_WORLD1=1 (a rehearsal knob for a one-GPU box: every all-reduce / broadcast /
    barrier of the sharded path is then really issued through RCCL with world_size 1, where it changes no value)."""
    if dist is None or not dist.is_initialized():
        return None
    floor = 1 if os.environ.get("L2HMC_COLLECTIVES_AT_WORLD1") == "1" else 2
    return dist if dist.get_world_size() >= floor else None


def shard_bounds(num_chains, world_size, rank):
    """Contiguous block [lo, hi) of chains owned by `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(int(num_chains), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_state(tensors, dist, src=0):
    """Replicate weights / masks / eps from `src` (what hvd.BroadcastGlobalVariablesHook(0)
    does in gauge_model.py:1008)."""
    if dist is None or not dist.is_initialized():
        return
    for t in tensors:
        dist.broadcast(t, src=src)


class StepStats:
    """Per-step fused scalar buffer [sum p_accept, sum |dQ|, n_chains], combined over the ranks by an asynchronous
    all-reduce(SUM) on a side stream so the next trajectory does not wait for it.  `reduce_every` steps share ONE
    all-reduce of their stacked buffers [k, 3] (every step's global sums are the same numbers as with one collective
    per step, k - 1 steps later).  Measured with a one-rank RCCL group at the headline shape
    (profiles/r04_world1_rccl_kernel_trace.txt, bench.py: config.world1_rccl): the event that hands a step's sums to the
    side stream drains the launch queue between two step kernels -- a 17.6 us gap, 1.2 % of a step, with one all-reduce
    per step; one such gap per 16 steps (0.3 %) with the grouping.  At N > 1 the collective's own kernel, which has to
    find a CU among workgroups that hold full register files, is divided by 16 too.  reduce_every = 1 restores one
    collective per step."""

    def __init__(self, device, dist=None, reduce_every=16):
        self.device = torch.device(device)
        self.dist = active(dist)
        self.reduce_every = max(1, int(reduce_every))
        self.total = torch.zeros(3, dtype=torch.float64)
        self._pending = []
        self._unreduced = []                       # step buffers waiting for their shared all-reduce
        self._dev_total = None                     # fp64 [3] on the device: steps folded since the last wait()
        self._count, self._count_n = None, -1
        self._side = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

    def push(self, p_accept, abs_dq):
        n = p_accept.numel()
        if self._count is None or self._count_n != n or self._count.device != p_accept.device:
            self._count = torch.full((1,), float(n), dtype=torch.float32, device=p_accept.device)   # once per batch size
            self._count_n = n
        buf = torch.cat([torch.stack((p_accept, abs_dq)).sum(dim=1, dtype=torch.float32), self._count])
        self.push_sums(buf)

    def push_sums(self, buf):
        """`buf` = device tensor [sum p_accept, sum |dQ|, n_chains] of one step (from the step kernel,
        l2hmc_gauge_mcmc_step_ex, or from push): queued; when sharded, every `reduce_every` steps go through one
        asynchronous all-reduce together."""
        if self.dist is None:
            self._pending.append((buf, None))
        else:
            self._unreduced.append(buf)
            if len(self._unreduced) >= self.reduce_every:
                self._reduce_unreduced()
        if len(self._pending) > 64:
            self._drain(keep=8)

    def _reduce_unreduced(self):
        """One all-reduce(SUM) of the queued steps' buffers, stacked [k, 3], on the side stream."""
        if not self._unreduced:
            return
        block = self._unreduced[0].reshape(1, -1) if len(self._unreduced) == 1 else torch.stack(self._unreduced)
        self._unreduced = []
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                work = self.dist.all_reduce(block, op=self.dist.ReduceOp.SUM, async_op=True)
            block.record_stream(self._side)
        else:
            work = self.dist.all_reduce(block, op=self.dist.ReduceOp.SUM, async_op=True)
        self._pending.append((block, work))

    def _drain(self, keep=0):
        """Fold all but the newest `keep` step buffers into the running device total: one stacked sum per drain (not
        one per step), no host synchronisation."""
        n = len(self._pending) - keep
        if n <= 0:
            return
        batch, self._pending = self._pending[:n], self._pending[n:]
        for _, work in batch:
            if work is not None:
                work.wait()
        # folded ON THE DEVICE: a device-to-host copy here would block the host until every step enqueued so far has
        # run -- the GPU then idles while the host catches up (measured: 0.8 ms per drain, 2.5 % of a 20-step region)
        part = torch.cat([b.detach().reshape(-1, 3) for b, _ in batch]).sum(dim=0, dtype=torch.float64)
        self._dev_total = part if self._dev_total is None else self._dev_total + part

    def join(self):
        """Device side only: the steps still queued get their all-reduce, and the current stream waits for every
        all-reduce issued so far (no host work, no copy); the folding of the step buffers into the host totals is left
        to wait() / the next drain."""
        if self.dist is not None:
            self._reduce_unreduced()
        if self._side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self._side)

    def wait(self):
        """Everything issued so far is folded into the host totals (blocks the host until those steps have run)."""
        self.join()
        self._drain(0)
        if self._dev_total is not None:
            self.total += self._dev_total.cpu()
            self._dev_total = None

    def mean_accept(self):
        self.wait()
        return float(self.total[0] / self.total[2]) if self.total[2] > 0 else float("nan")

    def mean_abs_dq(self):
        self.wait()
        return float(self.total[1] / self.total[2]) if self.total[2] > 0 else float("nan")
