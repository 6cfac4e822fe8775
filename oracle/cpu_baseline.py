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
    def __init__(self, T, X, num_steps, eps, masks, xnet, vnet, arch='generic'):
        self.T, self.X, self.N, self.arch = T, X, num_steps, arch
        self.eps = torch.tensor(float(eps), dtype=torch.float32)
        self.mask = torch.tensor(np.asarray(masks), dtype=torch.float32)
        self.xnet = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in xnet.items()}
        self.vnet = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in vnet.items()}

    def _front(self, p, a, which):
        """network/conv_net.py:251-262 for one input (channels_last), one torch op per Keras layer: Conv3D(F,(3,3,2),
        same, relu) -> MaxPool3D(2, s2, same) -> Conv3D(2F,(2,2,2), same, relu) -> MaxPool3D -> flatten."""
        import torch.nn.functional as Fn
        B = a.shape[0]
        h = a.reshape(B, 1, self.T, self.X, 2)
        w1 = p[f'conv_{which}1/W'].permute(4, 3, 0, 1, 2)           # Keras [kh,kw,kd,Cin,Cout] -> torch [Cout,Cin,kh,kw,kd]
        h = torch.relu(Fn.conv3d(Fn.pad(h, (0, 1, 1, 1, 1, 1)), w1, p[f'conv_{which}1/b']))
        h = Fn.max_pool3d(h, (2, 2, 2))
        w2 = p[f'conv_{which}2/W'].permute(4, 3, 0, 1, 2)
        h = torch.relu(Fn.conv3d(Fn.pad(h, (0, 1, 0, 1, 0, 1)), w2, p[f'conv_{which}2/b']))
        h = Fn.max_pool3d(h, (2, 2, 1))
        return h.permute(0, 2, 3, 4, 1).reshape(B, -1)

    def _net(self, p, inputs):
        v, x, t = inputs
        if self.arch == 'conv3D':
            v, x = self._front(p, v, 'v'), self._front(p, x, 'x')
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


class TorchCpuDynamics:
    """The generic integrator on a toy target (utils/dynamics.py:120-319, utils/sampler.py:28-59,
    utils/network.py:89-114, utils/distributions.py:32-39,151-158) at the reference's op granularity: one matmul
    per Linear, the energy gradient by autograd of the energy (`tf.gradients`, :241-242) -- including the
    reference's B x B `quadratic_gaussian` product (distributions.py:36-39) -- a Python loop over the leapfrog
    steps, both directions integrated."""

    def __init__(self, target, num_steps, eps, masks, xnet, vnet):
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)      # noqa: E731
        self.N = num_steps
        self.eps = torch.tensor(float(eps), dtype=torch.float32)
        self.mask = f32(masks)
        self.xnet = {k: f32(v) for k, v in xnet.items()}
        self.vnet = {k: f32(v) for k, v in vnet.items()}
        if hasattr(target, 'i_sigmas'):
            self.mus, self.precs = [f32(m) for m in target.mus], [f32(s) for s in target.i_sigmas]
            self.logc, self.gaussian = [float(np.log(c)) for c in target.constants], False
        else:
            self.mus, self.precs, self.logc, self.gaussian = [f32(target.mu)], [f32(target.i_sigma)], [0.0], True

    @staticmethod
    def _quad(x, mu, S):
        d = x - mu
        return 0.5 * torch.diagonal(d @ S @ d.t())                # distributions.py:36-39 (B x B)

    def _energy(self, x):
        if self.gaussian:
            return self._quad(x, self.mus[0], self.precs[0])
        V = torch.stack([-self._quad(x, m, S) + c for m, S, c in zip(self.mus, self.precs, self.logc)], dim=1)
        return -torch.logsumexp(V, dim=1)

    def _grad(self, x):
        x = x.detach().requires_grad_(True)
        g, = torch.autograd.grad(self._energy(x).sum(), x)
        return g

    @staticmethod
    def _net(p, inputs):
        a, b, t = inputs
        h = (a @ p['embed_1/W'] + p['embed_1/b']) + (b @ p['embed_2/W'] + p['embed_2/b']) \
            + (t @ p['embed_3/W'] + p['embed_3/b'])
        h = torch.relu(h)
        h = torch.relu(h @ p['linear_1/W'] + p['linear_1/b'])
        S = torch.exp(p['scale_s']) * torch.tanh(h @ p['linear_s/W'] + p['linear_s/b'])
        T = h @ p['linear_t/W'] + p['linear_t/b']
        F = torch.exp(p['scale_f']) * torch.tanh(h @ p['linear_f/W'] + p['linear_f/b'])
        return S, T, F

    def _step(self, x, v, step, bwd):
        eps = self.eps
        arg = 2 * np.pi * step / self.N
        t = torch.tensor([[np.cos(arg), np.sin(arg)]], dtype=torch.float32).repeat(x.shape[0], 1)
        m = self.mask[step]
        mb = 1. - m
        g = self._grad(x)
        S, T, F = self._net(self.vnet, [x, g, t])
        if not bwd:
            s1 = 0.5 * eps * S
            v = v * torch.exp(s1) + 0.5 * eps * (-(torch.exp(eps * F) * g) + T)
            S, T, F = self._net(self.xnet, [v, m * x, t])
            s2 = eps * S
            x = m * x + mb * (x * torch.exp(s2) + eps * (torch.exp(eps * F) * v + T))
            S, T, F = self._net(self.xnet, [v, mb * x, t])
            s3 = eps * S
            x = mb * x + m * (x * torch.exp(s3) + eps * (torch.exp(eps * F) * v + T))
            g = self._grad(x)
            S, T, F = self._net(self.vnet, [x, g, t])
            s4 = 0.5 * eps * S
            v = v * torch.exp(s4) + 0.5 * eps * (-(torch.exp(eps * F) * g) + T)
            return x, v, (s1 + s4 + mb * s2 + m * s3).sum(1)
        s1 = -0.5 * eps * S
        v = (v - 0.5 * eps * (-(torch.exp(eps * F) * g) + T)) * torch.exp(s1)
        S, T, F = self._net(self.xnet, [v, mb * x, t])
        s2 = -eps * S
        x = mb * x + m * (torch.exp(s2) * (x - eps * (torch.exp(eps * F) * v + T)))
        S, T, F = self._net(self.xnet, [v, m * x, t])
        s3 = -eps * S
        x = m * x + mb * (torch.exp(s3) * (x - eps * (torch.exp(eps * F) * v + T)))
        g = self._grad(x)
        S, T, F = self._net(self.vnet, [x, g, t])
        s4 = -0.5 * eps * S
        v = torch.exp(s4) * (v - 0.5 * eps * (-(torch.exp(eps * F) * g) + T))
        return x, v, (s1 + s4 + m * s2 + mb * s3).sum(1)

    def _kernel(self, x0, v0, bwd):
        x, v = x0, v0
        ld = torch.zeros(x.shape[0])
        for t in range(self.N):
            x, v, j = self._step(x, v, self.N - t - 1 if bwd else t, bwd)
            ld = ld + j
        h0 = self._energy(x0) + 0.5 * (v0 ** 2).sum(1)
        h1 = self._energy(x) + 0.5 * (v ** 2).sum(1)
        p = torch.exp(torch.minimum(h0 - h1 + ld, torch.zeros(())))
        return x, v, torch.where(torch.isfinite(p), p, torch.zeros_like(p))

    def propose(self, x, v0f, v0b, bits, u):
        """sampler.py:28-59 with do_mh_step=True."""
        xf, _, pf = self._kernel(x, v0f, False)
        xb, _, pb = self._kernel(x, v0b, True)
        m = bits.float()
        Lx = m[:, None] * xf + (1 - m)[:, None] * xb
        px = m * pf + (1 - m) * pb
        return Lx, px, torch.where(((px - u) >= 0)[:, None], Lx, x)


def time_cpu_toy_baseline(target, num_steps, eps, batch, xnet, vnet, masks, budget_s=10.0, min_calls=3):
    """`propose` of the toy configs (BASELINE.json configs[0], [1]) on the host cores; same thread probe and
    return shape as time_cpu_baseline."""
    dyn = TorchCpuDynamics(target, num_steps, eps, masks, xnet, vnet)
    g = torch.Generator().manual_seed(102)
    dim = int(np.asarray(masks).shape[1])
    x = torch.randn(batch, dim, generator=g)

    def one_call(xc):
        v0f, v0b = torch.randn(batch, dim, generator=g), torch.randn(batch, dim, generator=g)
        bits, u = torch.randint(0, 2, (batch,), generator=g), torch.rand(batch, generator=g)
        t0 = time.perf_counter()
        out = dyn.propose(xc, v0f, v0b, bits, u)
        return time.perf_counter() - t0, out[2]

    return _probe_and_time(one_call, x, batch * num_steps, budget_s, min_calls)


def _probe_and_time(one_call, x, units_per_call, budget_s, min_calls, threads=None):
    """threads=None: probe {1, 4, 8, ...} up to the CPU share for the best throughput; 'all': the whole share, no probe
    (for workloads where one call already takes seconds)."""
    cpus = effective_cpus()
    saved = torch.get_num_threads()
    cands = sorted({c for c in (1, 4, 8, 16, 32, 64, cpus) if c <= cpus} or {1})
    if threads is not None:
        cands = [cpus if threads == 'all' else int(threads)]
    best, best_t = cands[0], float("inf")
    try:
        for c in cands if len(cands) > 1 else ():   # probe: one warm-up + one timed call per candidate
            torch.set_num_threads(c)
            _, x = one_call(x)
            dt, x = one_call(x)
            if dt < best_t:
                best, best_t = c, dt
        torch.set_num_threads(best)
        if len(cands) == 1:
            _, x = one_call(x)                # warm-up
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
    return dict(value=units_per_call / med, cores=best, calls=len(times), seconds=med, cpus_available=cpus)


def time_cpu_baseline(T, X, num_steps, eps, beta, batch, xnet, vnet, masks, budget_s=15.0, min_calls=2,
                      arch='generic', threads=None):
    """Times apply_transition on a bounded sample: `batch` chains, repeated until ~budget_s of CPU work, with the
    thread count that serves this graph best on this host (probed over {4, 8, 16, 32, ...} up to the effective CPU
    share: the baseline is given its best configuration, not one thread per visible core).
    Returns dict(value=useful chain-LF/s, cores=threads used, calls, seconds, cpus_available)."""
    dyn = TorchCpuGaugeDynamics(T, X, num_steps, eps, masks, xnet, vnet, arch)
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

    return _probe_and_time(one_call, x, batch * num_steps, budget_s, min_calls, threads)
