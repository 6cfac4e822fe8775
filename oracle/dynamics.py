"""Oracle (test infrastructure): the generic L2HMC integrator, sampler glue and
toy targets (MoG / strongly-correlated Gaussian).

NumPy restatement of
  l2hmc/utils/dynamics.py:34-319     Dynamics (masks :85-103, time :105-110,
                                     _forward_step :120-170, _backward_step :172-225,
                                     energy/hamiltonian/grad_energy :227-242,
                                     forward/backward :255-310, p_accept :312-319)
  l2hmc/utils/sampler.py:28-59       propose, tf_accept
  l2hmc/utils/distributions.py:32-39 quadratic_gaussian, :56-80 Gaussian,
                                     :124-181 GMM, :231-243 gen_ring
The reference differentiates the energy with tf.gradients; the closed forms
below are checked against torch-CPU autograd in tests/test_oracle_kat.py.
"""
import numpy as np

from . import nets


# ------------------------------------------------------------- targets ----
def quadratic_gaussian(x, mu, S):
    """distributions.py:32-39: diag(0.5 (x-mu) S (x-mu)^T) -- evaluated row-wise
    here instead of through the reference's BxB product (same values)."""
    d = x - mu
    return 0.5 * np.einsum('bi,ij,bj->b', d, S, d)


class Gaussian:
    """distributions.py:56-80."""

    def __init__(self, mu, sigma):
        self.mu = np.asarray(mu, dtype=np.float64)
        self.sigma = np.asarray(sigma, dtype=np.float64)
        self.i_sigma = np.linalg.inv(np.copy(self.sigma))

    def energy(self, x):
        dt = x.dtype
        # the reference casts i_sigma and mu to float32 before use (:65-66)
        S = self.i_sigma.astype('float32').astype(dt)
        mu = self.mu.astype('float32').astype(dt)
        return quadratic_gaussian(x, mu, S)

    def grad_energy(self, x):
        dt = x.dtype
        S = self.i_sigma.astype('float32').astype(dt)
        mu = self.mu.astype('float32').astype(dt)
        return 0.5 * (x - mu) @ (S + S.T)

    def get_samples(self, n, rng):
        C = np.linalg.cholesky(self.sigma)
        return rng.standard_normal((n, self.sigma.shape[0])) @ C.T


class GMM:
    """distributions.py:124-181."""

    def __init__(self, mus, sigmas, pis):
        pis = np.array(pis, dtype=np.float64)
        if np.sum(pis) != 1.0:
            pis = pis / pis.sum()
        self.mus = [np.asarray(m, dtype=np.float64) for m in mus]
        self.sigmas = [np.asarray(s, dtype=np.float64) for s in sigmas]
        self.pis = pis
        self.nb_mixtures = len(pis)
        self.k = self.mus[0].shape[0]
        self.i_sigmas, self.constants = [], []
        for i, sigma in enumerate(self.sigmas):
            self.i_sigmas.append(np.linalg.inv(sigma).astype('float32'))
            det = np.sqrt((2 * np.pi) ** self.k * np.linalg.det(sigma)).astype('float32')
            self.constants.append((pis[i] / det).astype('float32'))

    def _V(self, x):
        dt = x.dtype
        cols = [-quadratic_gaussian(x, self.mus[i].astype('float32').astype(dt),
                                    self.i_sigmas[i].astype(dt))
                + np.log(self.constants[i]).astype(dt)
                for i in range(self.nb_mixtures)]
        return np.stack(cols, axis=1)

    def energy(self, x):
        """:151-158: -logsumexp_k(-quad_k + log c_k)."""
        V = self._V(x)
        m = np.max(V, axis=1, keepdims=True)
        return -(m[:, 0] + np.log(np.sum(np.exp(V - m), axis=1)))

    def grad_energy(self, x):
        dt = x.dtype
        V = self._V(x)
        m = np.max(V, axis=1, keepdims=True)
        w = np.exp(V - m)
        w = w / np.sum(w, axis=1, keepdims=True)
        g = np.zeros_like(x)
        for i in range(self.nb_mixtures):
            S = self.i_sigmas[i].astype(dt)
            mu = self.mus[i].astype('float32').astype(dt)
            g = g + w[:, i:i + 1] * (0.5 * (x - mu) @ (S + S.T))
        return g

    def get_samples(self, n, rng):
        cat = rng.choice(self.nb_mixtures, size=(n,), p=self.pis)
        out = np.empty((n, self.k))
        for k in range(self.nb_mixtures):
            sel = cat == k
            out[sel] = rng.multivariate_normal(self.mus[k], self.sigmas[k], size=int(sel.sum()))
        return out


def gen_ring(r=1.0, var=1.0, nb_mixtures=2):
    """distributions.py:231-243."""
    base = [np.array([r * np.cos(2 * np.pi * t / nb_mixtures),
                      r * np.sin(2 * np.pi * t / nb_mixtures)]) for t in range(nb_mixtures)]
    sigmas = [var * np.eye(2) for _ in range(nb_mixtures)]
    pis = [1. / nb_mixtures] * nb_mixtures
    pis[0] += 1 - sum(pis)
    return sigmas, GMM(base, sigmas, pis)


# ------------------------------------------------------------ dynamics ----
def make_masks(trajectory_length, x_dim, rng=None):
    """utils/dynamics.py:85-96 `_init_mask` (int(x_dim/2) ones per step)."""
    rng = rng if rng is not None else np.random
    out = []
    for _ in range(trajectory_length):
        ind = rng.permutation(np.arange(x_dim))[:int(x_dim / 2)]
        m = np.zeros((x_dim,))
        m[ind] = 1
        out.append(m)
    return np.stack(out)


class DynamicsOracle:
    def __init__(self, x_dim, target, trajectory_length, eps, masks,
                 xnet_params=None, vnet_params=None, hmc=False,
                 temperature=1.0, dtype=np.float64):
        self.x_dim = x_dim
        self.target = target
        self.trajectory_length = int(trajectory_length)
        self.dtype = dtype
        # quirk Q2: eps = exp(log(eps)) in the generic class (:51-60)
        self.eps = dtype(np.exp(np.log(dtype(eps))))
        self.mask = np.asarray(masks, dtype=dtype)
        self.hmc = hmc
        self.temperature = dtype(temperature)
        if hmc:
            self.XNet = nets.zero_net
            self.VNet = nets.zero_net
        else:
            xp, vp = nets.cast_params(xnet_params, dtype), nets.cast_params(vnet_params, dtype)
            self.XNet = lambda inp: nets.mlp_net(xp, inp)
            self.VNet = lambda inp: nets.mlp_net(vp, inp)

    def _get_mask(self, step):
        m = self.mask[int(step)]
        return m, 1. - m

    def _format_time(self, t, tile=1):
        arg = self.dtype(2 * np.pi) * self.dtype(t) / self.dtype(self.trajectory_length)
        tt = np.array([np.cos(arg), np.sin(arg)], dtype=self.dtype)
        return np.tile(tt[None, :], (tile, 1))

    def kinetic(self, v):
        return 0.5 * np.sum(np.square(v), axis=1)

    def energy(self, x):
        return self.target.energy(x) / self.temperature

    def hamiltonian(self, x, v):
        return self.energy(x) + self.kinetic(v)

    def grad_energy(self, x):
        return self.target.grad_energy(x) / self.temperature

    def _forward_step(self, x, v, step):
        """utils/dynamics.py:120-170."""
        eps = self.eps
        t = self._format_time(step, tile=x.shape[0])
        grad1 = self.grad_energy(x)
        S1 = self.VNet([x, grad1, t, None])
        sv1, tv1, fv1 = 0.5 * eps * S1[0], S1[1], eps * S1[2]
        v_h = v * np.exp(sv1) + 0.5 * eps * (-(np.exp(fv1) * grad1) + tv1)
        m, mb = self._get_mask(step)
        X1 = self.XNet([v_h, m * x, t, None])
        sx1, tx1, fx1 = eps * X1[0], X1[1], eps * X1[2]
        y = m * x + mb * (x * np.exp(sx1) + eps * (np.exp(fx1) * v_h + tx1))
        X2 = self.XNet([v_h, mb * y, t, None])
        sx2, tx2, fx2 = eps * X2[0], X2[1], eps * X2[2]
        x_o = mb * y + m * (y * np.exp(sx2) + eps * (np.exp(fx2) * v_h + tx2))
        grad2 = self.grad_energy(x_o)            # evaluated twice in the reference (Q7)
        S2 = self.VNet([x_o, grad2, t, None])
        sv2, tv2, fv2 = 0.5 * eps * S2[0], S2[1], eps * S2[2]
        v_o = v_h * np.exp(sv2) + 0.5 * eps * (-(np.exp(fv2) * grad2) + tv2)
        log_jac = np.sum(sv1 + sv2 + mb * sx1 + m * sx2, axis=1)
        return x_o, v_o, log_jac

    def _backward_step(self, x_o, v_o, step):
        """utils/dynamics.py:172-225."""
        eps = self.eps
        t = self._format_time(step, tile=x_o.shape[0])
        grad1 = self.grad_energy(x_o)
        S1 = self.VNet([x_o, grad1, t, None])
        sv2, tv2, fv2 = -0.5 * eps * S1[0], S1[1], eps * S1[2]
        v_h = (v_o - 0.5 * eps * (-(np.exp(fv2) * grad1) + tv2)) * np.exp(sv2)
        m, mb = self._get_mask(step)
        X1 = self.XNet([v_h, mb * x_o, t, None])
        sx2, tx2, fx2 = -eps * X1[0], X1[1], eps * X1[2]
        y = mb * x_o + m * (np.exp(sx2) * (x_o - eps * (np.exp(fx2) * v_h + tx2)))
        X2 = self.XNet([v_h, m * y, t, None])
        sx1, tx1, fx1 = -eps * X2[0], X2[1], eps * X2[2]
        x = m * y + mb * (np.exp(sx1) * (y - eps * (np.exp(fx1) * v_h + tx1)))
        grad2 = self.grad_energy(x)
        S2 = self.VNet([x, grad2, t, None])
        sv1, tv1, fv1 = -0.5 * eps * S2[0], S2[1], eps * S2[2]
        v = np.exp(sv1) * (v_h - 0.5 * eps * (-(np.exp(fv1) * grad2) + tv1))
        return x, v, np.sum(sv1 + sv2 + mb * sx1 + m * sx2, axis=1)

    def forward(self, x, init_v, log_jac=False):
        """:255-281; `init_v` replaces tf.random_normal when None in the reference."""
        x = np.asarray(x, dtype=self.dtype)
        v = np.asarray(init_v, dtype=self.dtype)
        X, V = x, v
        j = np.zeros((x.shape[0],), dtype=self.dtype)
        for t in range(self.trajectory_length):
            X, V, lj = self._forward_step(X, V, t)
            j = j + lj
        if log_jac:
            return X, V, j
        return X, V, self.p_accept(x, v, X, V, j)

    def backward(self, x, init_v, log_jac=False):
        """:283-310: step index trajectory_length - t - 1."""
        x = np.asarray(x, dtype=self.dtype)
        v = np.asarray(init_v, dtype=self.dtype)
        X, V = x, v
        j = np.zeros((x.shape[0],), dtype=self.dtype)
        for t in range(self.trajectory_length):
            X, V, lj = self._backward_step(X, V, self.trajectory_length - t - 1)
            j = j + lj
        if log_jac:
            return X, V, j
        return X, V, self.p_accept(x, v, X, V, j)

    def p_accept(self, x0, v0, x1, v1, log_jac):
        """:312-319."""
        e_new = self.hamiltonian(x1, v1)
        e_old = self.hamiltonian(x0, v0)
        with np.errstate(over='ignore', invalid='ignore'):
            p = np.exp(np.minimum(e_old - e_new + log_jac, 0.0))
        return np.where(np.isfinite(p), p, np.zeros_like(p))


def tf_accept(x, Lx, px, u):
    """sampler.py:57-59: accept iff px - u >= 0 (quirk Q5: non-strict)."""
    mask = (px - u) >= 0.
    return np.where(mask[:, None], Lx, x)


def propose(x, dynamics, v0_f, v0_b, dir_bits, u=None, do_mh_step=False):
    """sampler.py:28-55.  `dir_bits` in {0,1} stands in for
    tf.random_uniform(maxval=2, dtype=int32) (1 = forward); Lv is None because
    callers pass no init_v (quirk Q6) -- the mixed momentum is returned as a
    fifth value for kernel checks only."""
    x = np.asarray(x, dtype=dynamics.dtype)
    if dynamics.hmc:
        Lx, Lv, px = dynamics.forward(x, v0_f)
        return Lx, Lv, px, [tf_accept(x, Lx, px, u)], Lv
    mask = np.asarray(dir_bits, dtype=dynamics.dtype)[:, None]
    Lx1, Lv1, px1 = dynamics.forward(x, v0_f)
    Lx2, Lv2, px2 = dynamics.backward(x, v0_b)
    Lx = mask * Lx1 + (1 - mask) * Lx2
    Lv_mixed = mask * Lv1 + (1 - mask) * Lv2
    px = mask[:, 0] * px1 + (1 - mask)[:, 0] * px2
    outputs = []
    if do_mh_step:
        outputs.append(tf_accept(x, Lx, px, u))
    return Lx, None, px, outputs, Lv_mixed
