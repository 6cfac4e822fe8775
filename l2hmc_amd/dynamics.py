"""`Dynamics` with the reference's surface (l2hmc/utils/dynamics.py:34-319).

Two paths behind it, chosen per instance (`Dynamics.layered`):
  * one launch per trajectory (l2hmc_small_trajectory / l2hmc_small_propose): the packed toy targets of
    l2hmc_amd.distributions (GMM / Gaussian, x_dim <= 8, <= 8 components) with `network`-style nets of at most 64
    hidden units -- BASELINE configs 1 and 2;
  * layer by layer, for everything else the reference's constructor accepts (:35-43: ANY `energy_function`, any x_dim,
    `net_factory` of any `num_nodes`): per sub-update of utils/dynamics.py:120-225 one S/T/Q evaluation through
    l2hmc_stq_dense (any width) and one l2hmc_lf_update_v / _x launch; the energy gradient comes from the packed target's
    kernel or, for an arbitrary callable on torch tensors, from torch.autograd (the reference's tf.gradients, :241-242)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class Dynamics(object):
    def __init__(self, x_dim, energy_function, trajectory_length=10, eps=0.1, hmc=False, net_factory=None,
                 eps_trainable=True, use_temperature=False, device=None, seed=42):
        self.x_dim = x_dim
        self.use_temperature = use_temperature
        self.temperature = 1.0             # the reference feeds a placeholder (:48)
        self.first_layer_form = 0          # 0 by batch size; 1 / 2 force the toy kernel's matrix-pipe / VALU first layer
        self._device = device or torch.device("cuda", torch.cuda.current_device())
        # quirk Q2: eps = exp(alpha), alpha = log(eps) (:51-60)
        self.alpha = torch.log(torch.tensor(float(eps), dtype=torch.float32))
        self.eps_trainable = eps_trainable
        if not callable(energy_function):
            raise TypeError("energy_function must be callable: x [B, x_dim] -> energies [B]")
        self._fn = energy_function
        # packed toy target (l2hmc_amd.distributions) or None = an arbitrary callable on torch tensors
        self._target = getattr(energy_function, "target", None)
        if self._target is not None:
            if self._target.dim != x_dim:
                raise ValueError(f"x_dim={x_dim} but the target has dimension {self._target.dim}")
            self._target.to(self._device)
        self.trajectory_length = int(trajectory_length)
        self.hmc = hmc
        self._init_mask()
        if hmc:                             # :75-78
            self.XNet = lambda inp: [torch.zeros_like(inp[0]) for _ in range(3)]
            self.VNet = lambda inp: [torch.zeros_like(inp[0]) for _ in range(3)]
        else:
            self.XNet = net_factory(x_dim, scope='XNet', factor=2.0)
            self.VNet = net_factory(x_dim, scope='VNet', factor=1.0)
        self._seed, self._draws = int(seed), 0
        # the one-launch kernels hold x_dim <= 8 and 64 hidden units, and evaluate the packed targets only
        wide = (not hmc) and int(getattr(self.XNet, "num_nodes", 0)) > _lib.MAX_SMALL_NODES
        self.layered = self._target is None or int(x_dim) > _lib.MAX_SMALL_DIM or wide

    @property
    def eps(self):
        return torch.exp(self.alpha)

    def _init_mask(self):
        """:85-96 (legacy global NumPy stream)."""
        mask_per_step = []
        for _ in range(self.trajectory_length):
            ind = np.random.permutation(np.arange(self.x_dim))[:int(self.x_dim / 2)]
            m = np.zeros((self.x_dim,))
            m[ind] = 1
            mask_per_step.append(m)
        self.set_masks(np.stack(mask_per_step))

    def set_masks(self, masks):
        masks = np.asarray(masks, dtype=np.float32)
        if masks.shape != (self.trajectory_length, self.x_dim):
            raise ValueError(f"masks: expected {(self.trajectory_length, self.x_dim)}, got {masks.shape}")
        self.mask = _lib.as_dev(masks, self._device)

    def _get_mask(self, step):
        m = self.mask[int(step)]
        return m, 1. - m

    def _format_time(self, t, tile=1):
        """:105-110."""
        arg = np.float32(2 * np.pi) * np.float32(t) / np.float32(self.trajectory_length)
        return torch.tensor([[np.cos(arg), np.sin(arg)]], dtype=torch.float32).repeat(tile, 1)

    def _temp(self):
        return float(self.temperature) if self.use_temperature else 1.0

    def kinetic(self, v):
        v = _lib.as_dev(v, self._device)
        out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().l2hmc_kinetic_energy(v.data_ptr(), v.shape[0], v.shape[1], out.data_ptr(),
                                                   _lib.stream_ptr(self._device)))
        return out

    def energy(self, x, aux=None):
        """:227-236."""
        if self._target is None:
            return self._fn(_lib.as_dev(x, self._device).reshape(-1, self.x_dim)) / self._temp()
        return self._target.energy_grad(x, self._temp(), want_grad=False)[0]

    def hamiltonian(self, x, v, aux=None):
        return self.energy(x) + self.kinetic(v)

    def grad_energy(self, x, aux=None):
        """:241-242 -- closed form of the reference's tf.gradients for the packed targets, torch.autograd of the
        caller's function otherwise."""
        if self._target is None:
            with torch.enable_grad():
                xg = _lib.as_dev(x, self._device).reshape(-1, self.x_dim).detach().clone().requires_grad_(True)
                e = self._fn(xg)
                if e.shape != (xg.shape[0],):
                    raise ValueError(f"energy_function must return one energy per row: got {tuple(e.shape)}")
                (g,) = torch.autograd.grad(e.sum(), xg)
            return (g / self._temp()).contiguous()
        return self._target.energy_grad(x, self._temp())[1]

    # ---- layer-by-layer path (utils/dynamics.py:120-225 sub-update by sub-update) ------------------------------------
    def _sub_v(self, x, v, t, d, logdet):
        """:123-132 / :158-166 (d = 0), :175-185 / :213-223 (d = 1): half-kick with VNet([x, grad, t])."""
        g = self.grad_energy(x)
        S, T, Q = self.VNet([x, g, t])
        out, ld = torch.empty_like(v), torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().l2hmc_lf_update_v(v.data_ptr(), g.data_ptr(), S.data_ptr(), T.data_ptr(), Q.data_ptr(),
                                                float(self.eps), d, v.shape[0], self.x_dim, out.data_ptr(), ld.data_ptr(),
                                                _lib.stream_ptr(self._device)))
        logdet += ld
        return out

    def _sub_x(self, x, v, keep, t, d, logdet):
        """:134-156 (d = 0), :187-211 (d = 1): XNet([v, keep * x, t]); the kept coordinates pass through."""
        S, T, Q = self.XNet([v, keep * x, t])
        out, ld = torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().l2hmc_lf_update_x(x.data_ptr(), v.data_ptr(), keep.data_ptr(), S.data_ptr(), T.data_ptr(),
                                                Q.data_ptr(), float(self.eps), d, x.shape[0], self.x_dim, out.data_ptr(),
                                                ld.data_ptr(), _lib.stream_ptr(self._device)))
        logdet += ld
        return out

    def _layered_run(self, x, v, backward, log_jac):
        """forward (:255-281) or backward (:283-310) through the public sub-update operators."""
        x0, v0 = x, v
        lj = torch.zeros(x.shape[0], dtype=torch.float32, device=x.device)
        N = self.trajectory_length
        for i in range(N):
            step = N - i - 1 if backward else i
            t = self._format_time(step)
            m, mb = self._get_mask(step)
            if not backward:                                   # :120-170
                v = self._sub_v(x, v, t, 0, lj)
                x = self._sub_x(x, v, m, t, 0, lj)
                x = self._sub_x(x, v, mb, t, 0, lj)
                v = self._sub_v(x, v, t, 0, lj)
            else:                                              # :172-225
                v = self._sub_v(x, v, t, 1, lj)
                x = self._sub_x(x, v, mb, t, 1, lj)
                x = self._sub_x(x, v, m, t, 1, lj)
                v = self._sub_v(x, v, t, 1, lj)
        return (x, v, lj) if log_jac else (x, v, self.p_accept(x0, v0, x, v, lj))

    def _plan(self):
        if self.layered:
            raise NotImplementedError(
                "this Dynamics runs layer by layer (arbitrary energy function, x_dim > 8 or more than 64 hidden units): "
                "the one-launch kernels (l2hmc_small_*) and the one-launch training step do not hold it")
        p = _lib.SmallPlan(x_dim=self.x_dim, trajectory_length=self.trajectory_length, hmc=int(bool(self.hmc)),
                           eps=float(self.eps), first_layer_form=int(self.first_layer_form), masks=self.mask.data_ptr(),
                           target=self._target.struct(self._temp()), num_nodes=0)
        if not self.hmc:
            p.xnet, p.vnet = self.XNet.pack(), self.VNet.pack()
            p.num_nodes = p.xnet.H
        return p

    def _normal(self, shape):
        out = torch.empty(shape, dtype=torch.float32, device=self._device)
        _lib.check(_lib.lib().l2hmc_fill_normal(out.data_ptr(), out.numel(), self._seed, self._draws,
                                                _lib.stream_ptr(self._device)))
        self._draws += 1
        return out

    def _run(self, x, init_v, backward, log_jac):
        x = _lib.as_dev(x, self._device).reshape(-1, self.x_dim)
        v = _lib.as_dev(init_v, self._device) if init_v is not None else self._normal(tuple(x.shape))
        if self.layered:
            return self._layered_run(x, v, backward, log_jac)
        rows = x.shape[0]
        X, V = torch.empty_like(x), torch.empty_like(x)
        lj = torch.empty(rows, dtype=torch.float32, device=x.device)
        p = torch.empty_like(lj)
        dirs = torch.full((rows,), 1, dtype=torch.int32, device=x.device) if backward else None
        plan = self._plan()
        _lib.check(_lib.lib().l2hmc_small_trajectory(
            C.byref(plan), x.data_ptr(), _lib.dev_ptr(v, name="init_v"), _lib.dev_ptr(dirs, torch.int32), rows,
            X.data_ptr(), V.data_ptr(), lj.data_ptr(), p.data_ptr(), _lib.stream_ptr(self._device)))
        return (X, V, lj) if log_jac else (X, V, p)

    def both(self, x, init_v_forward=None, init_v_backward=None, log_jac=False):
        """forward(x) and backward(x) of the same chains in ONE launch (rows stacked, per-row direction):
        what `propose` needs (sampler.py:38-39).  Returns ((Xf, Vf, pf), (Xb, Vb, pb))."""
        x = _lib.as_dev(x, self._device).reshape(-1, self.x_dim)
        B = x.shape[0]
        vf = _lib.as_dev(init_v_forward, self._device) if init_v_forward is not None else self._normal(tuple(x.shape))
        vb = _lib.as_dev(init_v_backward, self._device) if init_v_backward is not None else self._normal(tuple(x.shape))
        if self.layered:
            return self._layered_run(x, vf, False, log_jac), self._layered_run(x, vb, True, log_jac)
        xx, vv = torch.cat([x, x]), torch.cat([vf, vb])
        dirs = torch.cat([torch.zeros(B, dtype=torch.int32, device=x.device),
                          torch.ones(B, dtype=torch.int32, device=x.device)])
        X, V = torch.empty_like(xx), torch.empty_like(xx)
        lj = torch.empty(2 * B, dtype=torch.float32, device=x.device)
        p = torch.empty_like(lj)
        plan = self._plan()
        _lib.check(_lib.lib().l2hmc_small_trajectory(
            C.byref(plan), xx.data_ptr(), vv.data_ptr(), dirs.data_ptr(), 2 * B, X.data_ptr(), V.data_ptr(),
            lj.data_ptr(), p.data_ptr(), _lib.stream_ptr(self._device)))
        third = lj if log_jac else p
        return (X[:B], V[:B], third[:B]), (X[B:], V[B:], third[B:])

    def forward(self, x, init_v=None, aux=None, log_path=False, log_jac=False):
        """:255-281."""
        if aux is not None:
            raise NotImplementedError("aux inputs are only used by the out-of-scope VAE scripts")
        return self._run(x, init_v, False, log_jac)

    def backward(self, x, init_v=None, aux=None, log_jac=False):
        """:283-310."""
        if aux is not None:
            raise NotImplementedError("aux inputs are only used by the out-of-scope VAE scripts")
        return self._run(x, init_v, True, log_jac)

    def p_accept(self, x0, v0, x1, v1, log_jac, aux=None):
        """:312-319."""
        e_new, e_old = self.hamiltonian(x1, v1), self.hamiltonian(x0, v0)
        lj = _lib.as_dev(log_jac, self._device)
        p = torch.empty_like(e_old)
        _lib.check(_lib.lib().l2hmc_accept_prob(e_old.data_ptr(), e_new.data_ptr(), lj.data_ptr(), p.numel(),
                                                p.data_ptr(), _lib.stream_ptr(self._device)))
        return p
