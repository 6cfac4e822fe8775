"""Toy targets with the reference's class surface (l2hmc/utils/distributions.py:
quadratic_gaussian :32-39, Gaussian :56-80, GMM :124-181, gen_ring :231-243).
Parameters live on the host as in the reference; `get_energy_function()` returns
a callable that evaluates on the device through l2hmc_mog_energy_grad and
carries the packed parameters (`.target`) so that Dynamics can hand them to the
fused trajectory kernel."""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib


class _PackedTarget:
    def __init__(self, mus, precs, log_consts, is_gaussian, device=None):
        self.K, self.dim = len(mus), int(np.asarray(mus[0]).shape[0])
        if self.dim > _lib.MAX_SMALL_DIM or self.K > _lib.MAX_MIX:
            raise ValueError(f"target: dim={self.dim} / K={self.K} beyond the fused kernel's limits "
                             f"({_lib.MAX_SMALL_DIM}, {_lib.MAX_MIX})")
        self.is_gaussian = int(is_gaussian)
        self._host = (np.stack([np.asarray(m, dtype=np.float32) for m in mus]),
                      np.stack([np.asarray(p, dtype=np.float32) for p in precs]),
                      np.asarray(log_consts, dtype=np.float32))
        self.device = None
        if device is not None:
            self.to(device)

    def to(self, device):
        """Parameters onto `device` (done on first use with the current CUDA device when nobody asked earlier)."""
        if self.device != device:
            self.device = device
            self.mu, self.prec, self.log_const = (_lib.as_dev(a, device) for a in self._host)
        return self

    def struct(self, temperature=1.0):
        if self.device is None:
            self.to(torch.device("cuda", torch.cuda.current_device()))
        return _lib.MogTarget(dim=self.dim, K=self.K, is_gaussian=self.is_gaussian,
                              temperature=float(temperature), mu=self.mu.data_ptr(),
                              prec=self.prec.data_ptr(), log_const=self.log_const.data_ptr())

    def energy_grad(self, x, temperature=1.0, want_grad=True):
        if self.device is None:
            self.to(torch.device("cuda", torch.cuda.current_device()))
        x = _lib.as_dev(x, self.device).reshape(-1, self.dim)
        e = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        g = torch.empty_like(x) if want_grad else None
        st = self.struct(temperature)
        _lib.check(_lib.lib().l2hmc_mog_energy_grad(C.byref(st), x.data_ptr(), x.shape[0], e.data_ptr(),
                                                    None if g is None else g.data_ptr(), _lib.stream_ptr(x.device)))
        return e, g


def _energy_fn(target):
    def fn(x, *args, **kwargs):
        return target.energy_grad(x, want_grad=False)[0]
    fn.target = target
    return fn


def quadratic_gaussian(x, mu, S):
    """:32-39 -- 0.5 (x-mu) S (x-mu)^T per row (the reference takes the diagonal of a BxB product)."""
    t = _PackedTarget([np.asarray(mu)], [np.asarray(S)], [0.], True)
    return t.energy_grad(x, want_grad=False)[0]


class Gaussian(object):
    """:56-80."""

    def __init__(self, mu, sigma):
        self.mu = np.asarray(mu)
        self.sigma = np.asarray(sigma)
        self.i_sigma = np.linalg.inv(np.copy(self.sigma))

    def get_energy_function(self):
        return _energy_fn(_PackedTarget([self.mu.astype('float32')], [self.i_sigma.astype('float32')], [0.], True))

    def get_samples(self, n):
        C_ = np.linalg.cholesky(self.sigma)
        X = np.random.randn(n, self.sigma.shape[0])
        return X.dot(C_.T)


class GMM(object):
    """:124-181."""

    def __init__(self, mus, sigmas, pis):
        assert len(mus) == len(sigmas)
        if not isinstance(pis, np.ndarray):
            pis = np.array(pis)
        if np.sum(pis) != 1.0:
            pis = pis / pis.sum()
        self.mus, self.sigmas, self.pis = mus, sigmas, pis
        self.nb_mixtures = len(pis)
        self.k = mus[0].shape[0]
        self.i_sigmas, self.constants = [], []
        for i, sigma in enumerate(sigmas):
            self.i_sigmas.append(np.linalg.inv(sigma).astype('float32'))
            det = np.sqrt((2 * np.pi) ** self.k * np.linalg.det(sigma)).astype('float32')
            self.constants.append((pis[i] / det).astype('float32'))

    def get_energy_function(self):
        return _energy_fn(_PackedTarget(self.mus, self.i_sigmas, np.log(np.asarray(self.constants)), False))

    def get_samples(self, n):
        categorical = np.random.choice(self.nb_mixtures, size=(n,), p=self.pis)
        counter_samples = collections.Counter(categorical)
        samples = [np.random.multivariate_normal(self.mus[k], self.sigmas[k], size=(v,))
                   for k, v in counter_samples.items()]
        samples = np.concatenate(samples, axis=0)
        np.random.shuffle(samples)
        return samples


def gen_ring(r=1.0, var=1.0, nb_mixtures=2):
    """:231-243."""
    base_points = [np.array([r * np.cos(2 * np.pi * t / nb_mixtures), r * np.sin(2 * np.pi * t / nb_mixtures)])
                   for t in range(nb_mixtures)]
    sigmas = [var * np.eye(2) for _ in range(nb_mixtures)]
    pis = [1. / nb_mixtures] * nb_mixtures
    pis[0] += 1 - sum(pis)
    return sigmas, GMM(base_points, sigmas, pis)
