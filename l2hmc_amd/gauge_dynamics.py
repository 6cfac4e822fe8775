"""`GaugeDynamics` with the reference's operator surface
(l2hmc/dynamics/gauge_dynamics.py:42-709), backed by the HIP kernels of
libl2hmc_hip.so.  `dynamics(position, beta)` returns
(position_post, momentum_post, accept_prob, position_out) exactly as
`apply_transition` (:195-259) does; the random draws the reference takes from
the TF graph seed (momenta :269, direction coin :223, MH uniform :246) are
generated on the device by the library's Philox stream, or injected through
keyword arguments so a caller (or a parity test) can replay given draws.

Tensors are contiguous fp32 CUDA tensors [batch, x_dim]; NumPy inputs are
copied to the current device.  There is no CPU path."""
import ctypes as C

import numpy as np
import numpy.random as npr
import torch

from . import _lib
from .network import ConvNet3D, GenericNet


class GaugeDynamics:
    """Dynamics engine of the L2HMC sampler on the 2D U(1) lattice."""

    def __init__(self, lattice, potential_fn, **kwargs):
        self.name = 'GaugeDynamics'
        self.lattice = lattice
        self.potential = potential_fn
        self.batch_size = self.lattice.samples.shape[0]
        self.x_dim = self.lattice.num_links
        # defaults of the reference's caller (gauge_model.py:596-603)
        self.hmc, self.network_arch, self.num_steps = False, 'generic', 5
        self.eps_trainable, self.data_format = True, 'channels_last'
        self.both_directions = True      # integrate fwd AND bwd like :211-218; False = selected only
        self.fused = True                # whole-trajectory kernel where the shape has one
        self.check_numerics = False      # True: raise on a non-finite trajectory like tf.check_numerics (:26-28)
        self.recompute = False           # True: layer-by-layer path forms every first-layer product anew (diagnostic)
        self.tiles16_only = False        # True: whole-step kernel on its 16-row form for every batch (A/B, bit-identity test)
        self.all_columns = False         # True: layer-by-layer path forms S/T/Q for every column of a position sub-update
        for key, val in kwargs.items():
            if key != 'eps':             # :73-75
                setattr(self, key, val)
        self._device = kwargs.get('device') or torch.device("cuda", torch.cuda.current_device())
        if getattr(potential_fn, "u1_lattice", None) is None:
            raise NotImplementedError(
                "the fused trajectory integrates the 2D U(1) action; pass lattice.get_energy_function()")
        self.eps = torch.tensor(float(kwargs.get('eps', 0.4)), dtype=torch.float32)   # :91-96
        self._construct_time()
        self._construct_masks_while()
        if self.hmc:                     # :102-108
            self.position_fn = lambda inp: [torch.zeros_like(inp[0]) for _ in range(3)]
            self.momentum_fn = lambda inp: [torch.zeros_like(inp[0]) for _ in range(3)]
        elif self.network_arch == 'generic':
            self._build_generic_nets()
        elif self.network_arch == 'conv3D':
            self._build_conv_nets_3D()
        elif self.network_arch == 'conv2D':
            raise NotImplementedError(
                "network_arch='conv2D' is out of scope (SURVEY.md section 2: not named by any benchmark "
                "config, and its 4-D outputs break the reference's own reduce_sum(axis=1))")
        else:                            # :117-119
            raise AttributeError("`self._network_arch` must be one of `'conv3D', 'conv2D', 'generic'.`")
        self._ws = _lib.Workspace()
        self._seed = int(kwargs.get('seed', 42))
        self._draws = 0

    # ---- construction ------------------------------------------------------
    def _build_generic_nets(self):
        """:169-187."""
        kwargs = {'x_dim': self.x_dim, 'links_shape': self.lattice.links.shape,
                  'num_hidden': int(4 * self.x_dim), 'name_scope': 'position', 'factor': 2.}
        self.position_fn = GenericNet(model_name='XNet', device=self._device, **kwargs)
        kwargs['factor'] = 1.
        kwargs['name_scope'] = 'momentum'
        self.momentum_fn = GenericNet(model_name='VNet', device=self._device, **kwargs)

    def _build_conv_nets_3D(self):
        """:121-143."""
        kwargs = {
            '_input_shape': (self.batch_size, *self.lattice.links.shape),
            'links_shape': self.lattice.links.shape,
            'x_dim': self.lattice.num_links,
            'factor': 2.,
            'spatial_size': self.lattice.space_size,
            'num_hidden': 2 * self.lattice.num_links,
            'num_filters': int(self.lattice.space_size),
            'filter_sizes': [(3, 3, 2), (2, 2, 2)],
            'name_scope': 'position',
            'data_format': self.data_format,
        }
        self.position_fn = ConvNet3D(model_name='XNet', device=self._device, **kwargs)
        kwargs['name_scope'] = 'momentum'
        kwargs['factor'] = 1.
        self.momentum_fn = ConvNet3D(model_name='VNet', device=self._device, **kwargs)

    def _construct_time(self):
        """:611-619."""
        self.ts = [torch.tensor([[np.cos(2 * np.pi * i / self.num_steps),
                                  np.sin(2 * np.pi * i / self.num_steps)]], dtype=torch.float32)
                   for i in range(self.num_steps)]

    def _construct_masks_while(self):
        """:651-661 -- legacy global NumPy stream, like the reference."""
        mask_per_step = []
        for _ in range(self.num_steps):
            idx = npr.permutation(np.arange(self.x_dim))[:self.x_dim // 2]
            mask = np.zeros((self.x_dim,))
            mask[idx] = 1
            mask_per_step.append(mask)
        self.set_masks(np.stack(mask_per_step))

    def set_masks(self, masks):
        masks = np.asarray(masks, dtype=np.float32)
        if masks.shape != (self.num_steps, self.x_dim):
            raise ValueError(f"masks: expected {(self.num_steps, self.x_dim)}, got {masks.shape}")
        self.mask = _lib.as_dev(masks, self._device)

    @property
    def variables(self):
        v = [self.eps]
        if not self.hmc:
            v += self.position_fn.variables + self.momentum_fn.variables
        return v

    @property
    def trainable_variables(self):
        v = [self.eps] if self.eps_trainable else []
        if not self.hmc:
            v += self.position_fn.trainable_variables + self.momentum_fn.trainable_variables
        return v

    # ---- plumbing ----------------------------------------------------------
    def _plan(self):
        p = _lib.GaugePlan(T=self.lattice.time_size, X=self.lattice.space_size, num_steps=self.num_steps,
                           hmc=int(bool(self.hmc)), eps=float(self.eps),
                           flags=(0 if self.fused else _lib.PLAN_LAYERED)
                           | (0 if self.both_directions else _lib.PLAN_SELECTED_ONLY)
                           | (_lib.PLAN_RECOMPUTE if self.recompute else 0)
                           | (_lib.PLAN_TILES16_ONLY if self.tiles16_only else 0)
                           | (_lib.PLAN_ALL_COLUMNS if self.all_columns else 0),
                           masks=_lib.dev_ptr(self.mask, name="mask"))
        if not self.hmc:
            p.xnet = self.position_fn.pack()
            p.vnet = self.momentum_fn.pack()
            if self.network_arch == 'conv3D':
                p.flags |= _lib.PLAN_CONV3D
                p.xfront = self.position_fn.pack_front()
                p.vfront = self.momentum_fn.pack_front()
        return p

    def _x(self, a):
        a = _lib.as_dev(a, self._device)
        return a.reshape(a.shape[0], -1)

    def _normal(self, shape):
        out = torch.empty(shape, dtype=torch.float32, device=self._device)
        _lib.check(_lib.lib().l2hmc_fill_normal(out.data_ptr(), out.numel(), self._seed, self._draws,
                                                _lib.stream_ptr(self._device)))
        self._draws += 1
        return out

    def _uniform(self, shape):
        out = torch.empty(shape, dtype=torch.float32, device=self._device)
        _lib.check(_lib.lib().l2hmc_fill_uniform(out.data_ptr(), out.numel(), self._seed, self._draws,
                                                 _lib.stream_ptr(self._device)))
        self._draws += 1
        return out

    def _dir(self, rows, backward):
        return torch.full((rows,), int(backward), dtype=torch.int32, device=self._device)

    # ---- public operator surface -------------------------------------------
    def __call__(self, position, beta, **draws):
        return self.apply_transition(position, beta, **draws)

    call = __call__

    def apply_transition(self, position, beta, momentum_f=None, momentum_b=None, coin=None, u=None):
        """:195-259 -> (position_post, momentum_post, accept_prob, position_out)."""
        x = self._x(position)
        B, D = x.shape
        if momentum_f is None and momentum_b is None and coin is None and u is None:
            # no draw injected: the library draws for itself (one stream pair from this object's counter, the
            # layout of the native MCMC step) -- one kernel launch where the plan has a whole-trajectory kernel
            x_prop, v_prop, x_out = (torch.empty_like(x) for _ in range(3))
            p = torch.empty(B, dtype=torch.float32, device=x.device)
            plan, L = self._plan(), _lib.lib()
            ws, nb = self._ws.get(L.l2hmc_gauge_mcmc_step_ws_bytes(C.byref(plan), B), x.device)
            draw, self._draws = _lib.step_draw_index(self._draws)
            _lib.check(L.l2hmc_gauge_transition_draw(
                C.byref(plan), float(beta), _lib.dev_ptr(x, name="position"), B, self._seed, draw, x_prop.data_ptr(),
                v_prop.data_ptr(), p.data_ptr(), x_out.data_ptr(), ws, nb, _lib.stream_ptr(self._device)))
            if self.check_numerics and not bool(torch.isfinite(x_prop).all() & torch.isfinite(v_prop).all()):
                raise FloatingPointError("check_numerics: non-finite value in the proposed configuration")
            return x_prop, v_prop, p, x_out
        v0f = self._x(momentum_f) if momentum_f is not None else self._normal((B, D))
        v0b = self._x(momentum_b) if momentum_b is not None else self._normal((B, D))
        coin = _lib.as_dev(coin, self._device) if coin is not None else self._uniform((B,))
        u = _lib.as_dev(u, self._device) if u is not None else self._uniform((B,))
        x_prop, v_prop, x_out = (torch.empty_like(x) for _ in range(3))
        p = torch.empty(B, dtype=torch.float32, device=x.device)
        plan, L = self._plan(), _lib.lib()
        both = int(bool(self.both_directions))
        ws, nb = self._ws.get(L.l2hmc_gauge_transition_ws_bytes(C.byref(plan), B, both), x.device)
        _lib.check(L.l2hmc_gauge_transition(
            C.byref(plan), float(beta), _lib.dev_ptr(x, name="position"), _lib.dev_ptr(v0f, name="momentum_f"),
            _lib.dev_ptr(v0b, name="momentum_b"), _lib.dev_ptr(coin, name="coin"), _lib.dev_ptr(u, name="u"),
            B, both, x_prop.data_ptr(), v_prop.data_ptr(), p.data_ptr(), x_out.data_ptr(), ws, nb,
            _lib.stream_ptr(self._device)))
        if self.check_numerics and not bool(torch.isfinite(x_prop).all() & torch.isfinite(v_prop).all()):
            # the reference wraps every exp of the sub-updates in tf.check_numerics and aborts the step;
            # the kernels propagate NaN / inf instead, and this opt-in check (one host sync) reports it
            raise FloatingPointError("check_numerics: non-finite value in the proposed configuration")
        return x_prop, v_prop, p, x_out

    def transition_kernel(self, position, beta, forward=True, momentum=None, return_logdet=False):
        """:261-313 -> (position_post, momentum_post, accept_prob)."""
        x = self._x(position)
        rows, D = x.shape
        v0 = self._x(momentum) if momentum is not None else self._normal((rows, D))
        x_out, v_out = torch.empty_like(x), torch.empty_like(x)
        sld = torch.empty(rows, dtype=torch.float32, device=x.device)
        p = torch.empty_like(sld)
        plan, L = self._plan(), _lib.lib()
        dirs = None if forward else self._dir(rows, True)
        ws, nb = self._ws.get(L.l2hmc_gauge_ws_bytes(C.byref(plan), rows), x.device)
        _lib.check(L.l2hmc_gauge_trajectory(
            C.byref(plan), float(beta), _lib.dev_ptr(x, name="position"), _lib.dev_ptr(v0, name="momentum"),
            _lib.dev_ptr(dirs, torch.int32), rows, x_out.data_ptr(), v_out.data_ptr(), sld.data_ptr(),
            p.data_ptr(), ws, nb, _lib.stream_ptr(self._device)))
        if return_logdet:
            return x_out, v_out, p, sld
        return x_out, v_out, p

    def _lf(self, position, momentum, beta, step, backward):
        x, v = self._x(position).clone(), self._x(momentum).clone()
        rows = x.shape[0]
        logdet = torch.zeros(rows, dtype=torch.float32, device=x.device)
        plan, L = self._plan(), _lib.lib()
        dirs = self._dir(rows, True) if backward else None
        ws, nb = self._ws.get(L.l2hmc_gauge_ws_bytes(C.byref(plan), rows), x.device)
        _lib.check(L.l2hmc_gauge_leapfrog(C.byref(plan), float(beta), int(step), x.data_ptr(), v.data_ptr(),
                                          _lib.dev_ptr(dirs, torch.int32), rows, logdet.data_ptr(), ws, nb,
                                          _lib.stream_ptr(self._device)))
        return x, v, logdet

    def _forward_lf(self, position, momentum, beta, step):
        """:412-445."""
        return self._lf(position, momentum, beta, step, backward=False)

    def _backward_lf(self, position, momentum, beta, step):
        """:447-483 -- the time/mask index is num_steps - step - 1, reversed inside as there."""
        return self._lf(position, momentum, beta, step, backward=True)

    # sub-updates on materialised S/T/Q (the fused trajectory never materialises them)
    def _stq(self, fn, a, b, t):
        return fn([a, b, t])

    def _update_momentum(self, position, momentum, beta, t, backward):
        x, v = self._x(position), self._x(momentum)
        grad = self.grad_potential(x, beta)
        S, T, Q = self._stq(self.momentum_fn, x, grad, t)           # :493-495
        v_out = torch.empty_like(v)
        logdet = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().l2hmc_lf_update_v(
            v.data_ptr(), grad.data_ptr(), _lib.dev_ptr(self._x(S)), _lib.dev_ptr(self._x(T)),
            _lib.dev_ptr(self._x(Q)), float(self.eps), int(backward), x.shape[0], x.shape[1],
            v_out.data_ptr(), logdet.data_ptr(), _lib.stream_ptr(self._device)))
        return v_out, logdet

    def _update_momentum_forward(self, position, momentum, beta, t):
        """:486-508."""
        return self._update_momentum(position, momentum, beta, t, False)

    def _update_momentum_backward(self, position, momentum, beta, t):
        """:537-561."""
        return self._update_momentum(position, momentum, beta, t, True)

    def _update_position(self, position, momentum, t, mask, mask_inv, backward):
        x, v = self._x(position), self._x(momentum)
        keep = _lib.as_dev(mask, self._device).reshape(-1)
        S, T, Q = self._stq(self.position_fn, v, keep[None, :] * x, t)   # :515-517
        x_out = torch.empty_like(x)
        logdet = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().l2hmc_lf_update_x(
            x.data_ptr(), v.data_ptr(), keep.data_ptr(), _lib.dev_ptr(self._x(S)), _lib.dev_ptr(self._x(T)),
            _lib.dev_ptr(self._x(Q)), float(self.eps), int(backward), x.shape[0], x.shape[1],
            x_out.data_ptr(), logdet.data_ptr(), _lib.stream_ptr(self._device)))
        return x_out, logdet

    def _update_position_forward(self, position, momentum, t, mask, mask_inv):
        """:511-534."""
        return self._update_position(position, momentum, t, mask, mask_inv, False)

    def _update_position_backward(self, position, momentum, t, mask, mask_inv):
        """:565-590."""
        return self._update_position(position, momentum, t, mask, mask_inv, True)

    def _compute_accept_prob(self, position, momentum, position_post, momentum_post, sumlogdet, beta):
        """:592-609."""
        old = self.hamiltonian(position, momentum, beta)
        new = self.hamiltonian(position_post, momentum_post, beta)
        sld = _lib.as_dev(sumlogdet, self._device)
        p = torch.empty_like(old)
        _lib.check(_lib.lib().l2hmc_accept_prob(old.data_ptr(), new.data_ptr(), sld.data_ptr(), old.numel(),
                                                p.data_ptr(), _lib.stream_ptr(self._device)))
        return p

    def _get_time(self, i):
        return self.ts[i]

    def _format_time(self, i, tile=1):
        """:625-633 -> [tile, 2] (cos, sin) of 2 pi i / num_steps in fp32."""
        arg = np.float32(2 * np.pi) * np.float32(i) / np.float32(self.num_steps)
        t = torch.tensor([[np.cos(arg), np.sin(arg)]], dtype=torch.float32)
        return t.repeat(tile, 1)

    def _get_mask_while(self, step):
        """:671-673."""
        m = self.mask[int(step)]
        return m, 1. - m

    def potential_energy(self, position, beta):
        """:675-681."""
        return float(beta) * self.potential(self._x(position))

    def kinetic_energy(self, v):
        """:683-689."""
        v = self._x(v)
        out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
        _lib.check(_lib.lib().l2hmc_kinetic_energy(v.data_ptr(), v.shape[0], v.shape[1], out.data_ptr(),
                                                   _lib.stream_ptr(self._device)))
        return out

    def hamiltonian(self, position, momentum, beta):
        """:691-696."""
        return self.potential_energy(position, beta) + self.kinetic_energy(momentum)

    def grad_potential(self, position, beta, check_numerics=True):
        """:698-709 -- closed form of the autodiff the reference runs."""
        return self.lattice.grad_action(self._x(position), beta)
