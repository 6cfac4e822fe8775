"""Host-side mirror of the reference's `GaugeLattice` for link_type 'U1'
(l2hmc/lattice/lattice.py:61-362).  Storage (links / samples as NumPy arrays,
shape [T, X, 2] / [B, T, X, 2]) and attribute names follow the reference; the
batched action / force / observables run in the HIP kernel of
csrc/u1_lattice.hip on device tensors."""
import numpy as np
import torch
from scipy.special import i0, i1

from . import _lib


def u1_plaq_exact(beta):
    """lattice.py:31-33."""
    return i1(beta) / i0(beta)


def _x2d(x, D):
    x = _lib.as_dev(x)
    return x.reshape(-1, D)


def u1_observables(x, time_size, space_size, beta=1.0, want_force=False):
    """One pass over x: [rows, 2*T*X] -> dict(action, avg_plaq, top_charge[, force=beta*dS/dx])."""
    D = 2 * time_size * space_size
    x = _x2d(x, D)
    rows = x.shape[0]
    out = {k: torch.empty(rows, dtype=torch.float32, device=x.device)
           for k in ("action", "avg_plaq", "top_charge")}
    force = torch.empty_like(x) if want_force else None
    _lib.check(_lib.lib().l2hmc_u1_action_force(
        _lib.dev_ptr(x, name="x"), rows, time_size, space_size, float(beta), out["action"].data_ptr(),
        None if force is None else force.data_ptr(), out["avg_plaq"].data_ptr(),
        out["top_charge"].data_ptr(), _lib.stream_ptr(x.device)))
    if want_force:
        out["force"] = force
    return out


class GaugeLattice(object):
    """lattice.py:61-162 (U1 only; SU(2)/SU(3) operators are out of scope)."""

    def __init__(self, time_size, space_size, dim, link_type, num_samples=None, rand=False):
        if link_type.upper() != 'U1':
            raise NotImplementedError("only link_type='U1' is on the MI355X hot path (SURVEY.md section 2)")
        if dim != 2:
            raise NotImplementedError("the reference's batched action is 2-D (lattice.py:337-362)")
        self.time_size, self.space_size, self.dim, self.link_type = time_size, space_size, dim, link_type
        self.link_shape = ()
        sites_shape = (time_size, space_size)
        links_shape = (time_size, space_size, dim)
        self.sites = np.zeros(sites_shape, dtype=np.float32)
        self.links = np.zeros(links_shape, dtype=np.float32)
        if rand:   # lattice.py:131-135 (legacy global stream, as the reference)
            self.links = np.array(np.random.uniform(0, 2 * np.pi, links_shape), dtype=np.float32)
        self.site_idxs = self.sites.shape
        self.link_idxs = self.links.shape
        self.num_sites = int(np.cumprod(self.sites.shape)[-1])
        self.num_links = self.num_sites * self.dim
        self.num_plaquettes = self.time_size * self.space_size
        self.bases = np.eye(self.dim, dtype=int)
        if num_samples:
            self.num_samples = num_samples
            self.samples = self.get_links_samples(num_samples, rand=rand)
            self.samples[0] = self.links

    def _generate_links(self, rand=False, link_type=None):
        if rand:
            return 2 * np.pi * np.random.rand(*self.links.shape)
        return np.zeros(self.links.shape)

    def get_links_samples(self, num_samples, rand=False, link_type=None):
        return np.array([self._generate_links(rand, link_type) for _ in range(num_samples)])

    # ---- batched ops on the device (lattice.py:274-362)
    def get_energy_function(self, samples=None):
        def fn(samples):
            return self.total_action(samples)
        fn.u1_lattice = self     # lets GaugeDynamics recognise the fused target
        return fn

    def total_action(self, samples):
        return u1_observables(samples, self.time_size, self.space_size)["action"]

    def calc_plaq_observables(self, samples):
        o = u1_observables(samples, self.time_size, self.space_size)
        return o["action"], o["avg_plaq"], torch.floor(0.1 + o["top_charge"])   # lattice.py:309-311

    def calc_plaq_sums(self, samples):
        """gauge_model.py:659-681 -> [B, T, X]."""
        x = _x2d(samples, self.num_links)
        out = torch.empty(x.shape[0], self.time_size, self.space_size, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().l2hmc_u1_plaq_sums(_lib.dev_ptr(x, name="x"), x.shape[0], self.time_size,
                                                 self.space_size, out.data_ptr(), _lib.stream_ptr(x.device)))
        return out

    def grad_action(self, samples, beta=1.0):
        return u1_observables(samples, self.time_size, self.space_size, beta, want_force=True)["force"]
