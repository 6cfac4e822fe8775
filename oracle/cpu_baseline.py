"""Oracle-side CPU baseline (test / bench infrastructure, never shipped).

An op-for-op torch-CPU fp32 port of the reference's TF graph for one
`apply_transition` (dynamics/gauge_dynamics.py:195-313, :412-609) at the
reference's op granularity: one matmul per Dense layer (generic_net.py:129-146),
the force by autograd of the roll/cos action (gauge_dynamics.py:698-709 over
lattice.py:337-362), a Python loop standing in for tf.while_loop, BOTH
directions integrated, nothing fused.  It exists only so bench.py can time "the
reference's CPU path" on the GPU box's host cores (TensorFlow itself is not
installable here: SURVEY.md 8c/8d); it is labelled kind="port".
"""
import time

import numpy as np
import torch


class TorchCpuGaugeDynamics:
    def __init__(self, T, X, num_steps, eps, masks, xnet, vnet):
        self.T, self.X, self.N = T, X, num_steps
        self.eps = torch.tensor(float(eps), dtype=torch.float32)
        self.mask = torch.tensor(np.asarray(masks), dtype=torch.float32)
        self.xnet = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in xnet.items()}
        self.vnet = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in vnet.items()}

    @staticmethod
    def _net(p, inputs):
        v, x, t = inputs
        h = (v @ p['v_layer/W'] + p['v_layer/b']) + (x @ p['x_layer/W'] + p['x_layer/b']) \
            + (t @ p['t_layer/W'] + p['t_layer/b'])
        h = torch.relu(h)
        h = torch.relu(h @ p['h_layer/W'] + p['h_layer/b'])
        scale = torch.tanh(h @ p['scale_layer/W'] + p['scale_layer/b']) * torch.exp(p['coeff_scale'])
        translation = h @ p['translation_layer/W'] + p['translation_layer/b']
        transformation = (h @ p['transformation_layer/W'] + p['transformation_layer/b']) \
            * torch.exp(p['coeff_transformation'])
        return scale, translation, transformation

    def _action(self, x):
        s = x.reshape(x.shape[0], self.T, self.X, 2)
        P = s[..., 0] - s[..., 1] - torch.roll(s[..., 0], -1, 2) + torch.roll(s[..., 1], -1, 1)
        return torch.sum(1. - torch.cos(P), dim=(1, 2))

    def _grad(self, x, beta):
        x = x.detach().requires_grad_(True)
        g, = torch.autograd.grad((beta * self._action(x)).sum(), x)
        return g

    def _time(self, i, B):
        arg = 2 * np.pi * i / self.N
        return torch.tensor([[np.cos(arg), np.sin(arg)]], dtype=torch.float32).repeat(B, 1)

    def _upd_v(self, x, v, beta, t, bwd):
        g = self._grad(x, beta)
        S, T, Q = self._net(self.vnet, [x, g, t])
        eps = self.eps
        if not bwd:
            S = S * (0.5 * eps)
            return v * torch.exp(S) - 0.5 * eps * (torch.exp(Q * eps) * g - T), S.sum(1)
        S = S * (-0.5 * eps)
        return torch.exp(S) * (v + 0.5 * eps * (torch.exp(Q * eps) * g - T)), S.sum(1)

    def _upd_x(self, x, v, t, m, mi, bwd):
        S, T, Q = self._net(self.xnet, [v, m * x, t])
        eps = self.eps
        if not bwd:
            S = S * eps
            tmp = x * torch.exp(S) + eps * (torch.exp(Q * eps) * v + T)
        else:
            S = S * (-eps)
            tmp = torch.exp(S) * (x - eps * (torch.exp(Q * eps) * v + T))
        return m * x + mi * tmp, (mi * S).sum(1)

    def _lf(self, x, v, beta, step, bwd):
        i = self.N - step - 1 if bwd else step
        t = self._time(i, x.shape[0])
        m = self.mask[i]
        mi = 1. - m
        v, l1 = self._upd_v(x, v, beta, t, bwd)
        if not bwd:
            x, l2 = self._upd_x(x, v, t, m, mi, bwd)
            x, l3 = self._upd_x(x, v, t, mi, m, bwd)
        else:
            x, l2 = self._upd_x(x, v, t, mi, m, bwd)
            x, l3 = self._upd_x(x, v, t, m, mi, bwd)
        v, l4 = self._upd_v(x, v, beta, t, bwd)
        return x, v, l1 + l2 + l3 + l4

    def _kernel(self, x0, v0, beta, bwd):
        x, v = x0, v0
        ld = torch.zeros(x.shape[0])
        for step in range(self.N):
            x, v, j = self._lf(x, v, beta, step, bwd)
            ld = ld + j
        h0 = beta * self._action(x0) + 0.5 * (v0 ** 2).sum(1)
        h1 = beta * self._action(x) + 0.5 * (v ** 2).sum(1)
        p = torch.exp(torch.minimum(h0 - h1 + ld, torch.zeros(())))
        return x, v, torch.where(torch.isfinite(p), p, torch.zeros_like(p))

    def apply_transition(self, x, beta, v0f, v0b, coin, u):
        xf, vf, pf = self._kernel(x, v0f, beta, False)
        xb, vb, pb = self._kernel(x, v0b, beta, True)
        fm = (coin > 0.5).float()
        bm = 1. - fm
        xp = fm[:, None] * xf + bm[:, None] * xb
        vp = fm[:, None] * vf + bm[:, None] * vb
        p = fm * pf + bm * pb
        am = (p > u).float()
        return xp, vp, p, am[:, None] * xp + (1. - am)[:, None] * x


def effective_cpus():
    """CPUs this process may actually use: the cgroup quota (v2 cpu.max / v1 cfs_quota) if there is one, else the
    affinity mask.  Oversubscribing a quota with one thread per visible core makes torch-CPU several times slower
    (measured: 128 threads on a 16-CPU share ran this graph 6x slower than 8 threads)."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def time_cpu_baseline(T, X, num_steps, eps, beta, batch, xnet, vnet, masks, budget_s=15.0, min_calls=2):
    """Times apply_transition on a bounded sample: `batch` chains, repeated until ~budget_s of CPU work, with the
    thread count that serves this graph best on this host (probed over {4, 8, 16, 32, ...} up to the effective CPU
    share: the baseline is given its best configuration, not one thread per visible core).
    Returns dict(value=useful chain-LF/s, cores=threads used, calls, seconds, cpus_available)."""
    dyn = TorchCpuGaugeDynamics(T, X, num_steps, eps, masks, xnet, vnet)
    g = torch.Generator().manual_seed(103)
    D = 2 * T * X
    x = torch.rand(batch, D, generator=g) * (2 * np.pi)

    def one_call(xc):
        v0f = torch.randn(batch, D, generator=g)
        v0b = torch.randn(batch, D, generator=g)
        coin = torch.rand(batch, generator=g)
        u = torch.rand(batch, generator=g)
        t0 = time.perf_counter()
        out = dyn.apply_transition(xc, beta, v0f, v0b, coin, u)
        return time.perf_counter() - t0, torch.remainder(out[3], 2 * np.pi)

    cpus = effective_cpus()
    saved = torch.get_num_threads()
    cands = sorted({c for c in (4, 8, 16, 32, 64, cpus) if c <= cpus} or {1})
    best, best_t = cands[0], float("inf")
    try:
        for c in cands:                       # probe: one warm-up + one timed call per candidate
            torch.set_num_threads(c)
            _, x = one_call(x)
            dt, x = one_call(x)
            if dt < best_t:
                best, best_t = c, dt
        torch.set_num_threads(best)
        times = []
        t_start = time.perf_counter()
        while True:
            dt, x = one_call(x)
            times.append(dt)
            if len(times) >= min_calls and time.perf_counter() - t_start > budget_s:
                break
    finally:
        torch.set_num_threads(saved)
    med = float(np.median(times))
    return dict(value=batch * num_steps / med, cores=best, calls=len(times), seconds=med, cpus_available=cpus)
