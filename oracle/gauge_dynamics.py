"""Oracle (test infrastructure): the lattice L2HMC integrator.

NumPy restatement of l2hmc/dynamics/gauge_dynamics.py:
  :195-259  apply_transition       :261-313  transition_kernel
  :412-445  _forward_lf            :447-483  _backward_lf
  :486-508  _update_momentum_forward   :537-561 _update_momentum_backward
  :511-534  _update_position_forward   :565-590 _update_position_backward
  :592-609  _compute_accept_prob   :625-633  _format_time
  :651-673  masks                  :675-709  energies / grad_potential

All randomness (initial momenta, direction coin, MH uniform, masks, weights)
is an explicit argument; the reference draws it from the TF graph seed.
`dtype` selects the arithmetic (np.float64 = checker, np.float32 = the
reference's own precision, same op order).
"""
import numpy as np

from . import lattice as lat
from . import nets


def make_masks(num_steps, x_dim, rng=None):
    """gauge_dynamics.py:651-661 `_construct_masks_while`: per LF step a 0/1
    vector with exactly x_dim//2 ones at permutation(x_dim)[:x_dim//2], drawn
    from the legacy global NumPy stream (np.random.seed(42) in
    gauge_model.py:62,195).  Pass a RandomState for a private stream."""
    rng = rng if rng is not None else np.random
    out = []
    for _ in range(num_steps):
        idx = rng.permutation(np.arange(x_dim))[:x_dim // 2]
        m = np.zeros((x_dim,))
        m[idx] = 1
        out.append(m)
    return np.stack(out)


class GaugeDynamicsOracle:
    """Same method names as the reference class so tests read alike."""

    def __init__(self, time_size, space_size, num_steps, eps, masks,
                 xnet_params=None, vnet_params=None, network_arch='generic',
                 hmc=False, dtype=np.float64):
        self.T, self.X = time_size, space_size
        self.links_shape = (time_size, space_size, 2)
        self.x_dim = 2 * time_size * space_size
        self.num_steps = int(num_steps)
        self.dtype = dtype
        self.eps = dtype(eps)
        self.mask = np.asarray(masks, dtype=dtype)
        self.hmc = hmc
        self.network_arch = network_arch
        if hmc:
            self.position_fn = nets.zero_net
            self.momentum_fn = nets.zero_net
        elif network_arch == 'generic':
            xp, vp = nets.cast_params(xnet_params, dtype), nets.cast_params(vnet_params, dtype)
            self.position_fn = lambda inp: nets.generic_net(xp, inp)
            self.momentum_fn = lambda inp: nets.generic_net(vp, inp)
        elif network_arch == 'conv3D':
            xp, vp = nets.cast_params(xnet_params, dtype), nets.cast_params(vnet_params, dtype)
            self.position_fn = lambda inp: nets.conv3d_net(xp, inp, self.links_shape)
            self.momentum_fn = lambda inp: nets.conv3d_net(vp, inp, self.links_shape)
        else:
            # gauge_dynamics.py:117-119
            raise AttributeError("`network_arch` must be one of 'conv3D', 'generic'.")

    # ---- energies: gauge_dynamics.py:675-709 ----
    def potential_energy(self, position, beta):
        return self.dtype(beta) * lat.total_action(position, self.T, self.X)

    def kinetic_energy(self, v):
        return 0.5 * np.sum(v ** 2, axis=1)

    def hamiltonian(self, position, momentum, beta):
        return self.potential_energy(position, beta) + self.kinetic_energy(momentum)

    def grad_potential(self, position, beta):
        return self.dtype(beta) * lat.grad_action(position, self.T, self.X)

    # ---- time / masks: :625-633, :671-673 ----
    def _format_time(self, i, tile=1):
        two_pi = self.dtype(2 * np.pi)
        arg = two_pi * self.dtype(i) / self.dtype(self.num_steps)
        t = np.array([np.cos(arg), np.sin(arg)], dtype=self.dtype)
        return np.tile(t[None, :], (tile, 1))

    def _get_mask_while(self, step):
        m = self.mask[int(step)]
        return m, 1. - m

    # ---- sub-updates ----
    def _update_momentum_forward(self, position, momentum, beta, t):
        grad = self.grad_potential(position, beta)
        scale, translation, transformed = self.momentum_fn([position, grad, t])
        scale = scale * (0.5 * self.eps)
        transformed = transformed * self.eps
        momentum = (momentum * np.exp(scale)
                    - 0.5 * self.eps * (np.exp(transformed) * grad - translation))
        return momentum, np.sum(scale, axis=1)

    def _update_position_forward(self, position, momentum, t, mask, mask_inv):
        scale, translation, transformed = self.position_fn([momentum, mask * position, t])
        scale = scale * self.eps
        transformed = transformed * self.eps
        tmp = position * np.exp(scale) + self.eps * (np.exp(transformed) * momentum + translation)
        position = mask * position + mask_inv * tmp
        return position, np.sum(mask_inv * scale, axis=1)

    def _update_momentum_backward(self, position, momentum, beta, t):
        grad = self.grad_potential(position, beta)
        scale, translation, transformed = self.momentum_fn([position, grad, t])
        scale = scale * (-0.5 * self.eps)
        transformed = transformed * self.eps
        momentum = np.exp(scale) * (momentum + 0.5 * self.eps
                                    * (np.exp(transformed) * grad - translation))
        return momentum, np.sum(scale, axis=1)

    def _update_position_backward(self, position, momentum, t, mask, mask_inv):
        scale, translation, transformed = self.position_fn([momentum, mask * position, t])
        scale = scale * (-self.eps)
        transformed = transformed * self.eps
        tmp = np.exp(scale) * (position - self.eps * (np.exp(transformed) * momentum + translation))
        position = position * mask + mask_inv * tmp
        return position, np.sum(mask_inv * scale, axis=1)

    # ---- one augmented leapfrog step: :412-483 ----
    def _forward_lf(self, position, momentum, beta, step):
        t = self._format_time(step, tile=position.shape[0])
        mask, mask_inv = self._get_mask_while(step)
        sumlogdet = 0.
        momentum, logdet = self._update_momentum_forward(position, momentum, beta, t)
        sumlogdet = sumlogdet + logdet
        position, logdet = self._update_position_forward(position, momentum, t, mask, mask_inv)
        sumlogdet = sumlogdet + logdet
        position, logdet = self._update_position_forward(position, momentum, t, mask_inv, mask)
        sumlogdet = sumlogdet + logdet
        momentum, logdet = self._update_momentum_forward(position, momentum, beta, t)
        sumlogdet = sumlogdet + logdet
        return position, momentum, sumlogdet

    def _backward_lf(self, position, momentum, beta, step):
        rstep = self.num_steps - step - 1           # :453-457 index reversal INSIDE
        t = self._format_time(rstep, tile=position.shape[0])
        mask, mask_inv = self._get_mask_while(rstep)
        sumlogdet = 0.
        momentum, logdet = self._update_momentum_backward(position, momentum, beta, t)
        sumlogdet = sumlogdet + logdet
        position, logdet = self._update_position_backward(position, momentum, t, mask_inv, mask)
        sumlogdet = sumlogdet + logdet
        position, logdet = self._update_position_backward(position, momentum, t, mask, mask_inv)
        sumlogdet = sumlogdet + logdet
        momentum, logdet = self._update_momentum_backward(position, momentum, beta, t)
        sumlogdet = sumlogdet + logdet
        return position, momentum, sumlogdet

    # ---- trajectory: :261-313 ----
    def transition_kernel(self, position, beta, momentum, forward=True, trace=None):
        """`momentum` stands in for tf.random_normal(tf.shape(position)) (:269)."""
        lf_fn = self._forward_lf if forward else self._backward_lf
        position = np.asarray(position, dtype=self.dtype)
        momentum = np.asarray(momentum, dtype=self.dtype)
        x, v = position, momentum
        logdet = np.zeros((position.shape[0],), dtype=self.dtype)
        for step in range(self.num_steps):
            x, v, j = lf_fn(x, v, beta, step)
            logdet = logdet + j
            if trace is not None:
                trace.append((x.copy(), v.copy(), logdet.copy()))
        p = self._compute_accept_prob(position, momentum, x, v, logdet, beta)
        return x, v, p, logdet

    def _compute_accept_prob(self, position, momentum, position_post, momentum_post,
                             sumlogdet, beta):
        """:592-609: exp(min(H_old - H_new + sumlogdet, 0)); non-finite -> 0."""
        old = self.hamiltonian(position, momentum, beta)
        new = self.hamiltonian(position_post, momentum_post, beta)
        with np.errstate(over='ignore', invalid='ignore'):
            prob = np.exp(np.minimum(old - new + sumlogdet, 0.))
        return np.where(np.isfinite(prob), prob, np.zeros_like(prob))

    # ---- full MCMC step: :195-259 ----
    def apply_transition(self, position, beta, v0_f, v0_b, coin, u):
        """coin ~ U(0,1) picks forward iff coin > 0.5 (:221-227); accept iff
        p > u, strict (:245-250, quirk Q5)."""
        position = np.asarray(position, dtype=self.dtype)
        xf, vf, pf, _ = self.transition_kernel(position, beta, v0_f, forward=True)
        xb, vb, pb, _ = self.transition_kernel(position, beta, v0_b, forward=False)
        fm = (np.asarray(coin) > 0.5).astype(self.dtype)
        bm = 1. - fm
        x_post = fm[:, None] * xf + bm[:, None] * xb
        v_post = fm[:, None] * vf + bm[:, None] * vb
        p = fm * pf + bm * pb
        am = (p > np.asarray(u, dtype=self.dtype)).astype(self.dtype)
        x_out = am[:, None] * x_post + (1. - am)[:, None] * position
        return x_post, v_post, p, x_out
