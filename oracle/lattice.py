"""Oracle (test infrastructure): 2D U(1) lattice action, force and observables.

Restates, in NumPy, the batched graph ops of
  l2hmc/lattice/lattice.py:31-33     u1_plaq_exact
  l2hmc/lattice/lattice.py:47-49     project_angle
  l2hmc/lattice/lattice.py:285-313   calc_plaq_observables
  l2hmc/lattice/lattice.py:337-362   total_action
  l2hmc/gauge_model.py:78-108        project_angle / project_angle_approx
  l2hmc/gauge_model.py:659-725       _calc_plaq_sums/_total_actions/_avg_plaqs/_top_charges(_diff)
and the analytic force written out in
  l2hmc/lattice/gauge_lattice.py:427-459,485-489
which the live path obtains by autodiff (dynamics/gauge_dynamics.py:698-709).

Layout: one chain is the row-major flattening of links[t][x][mu], mu fastest
(lattice.py:121-129, gauge_model.py:1131-1134), so D = 2*T*X.
"""
import numpy as np
from scipy.special import i0, i1

TWO_PI = 2.0 * np.pi


def u1_plaq_exact(beta):
    """lattice.py:31-33 -- exact <avg plaquette> of 2D U(1)."""
    return i1(beta) / i0(beta)


def project_angle(a):
    """lattice.py:47-49 / gauge_model.py:78-80: a - 2pi*floor((a+pi)/2pi)."""
    return a - TWO_PI * np.floor((a + np.pi) / TWO_PI)


def project_angle_approx(a, N=5):
    """gauge_model.py:94-108: truncated Fourier series of the sawtooth,
    sum_{n=1}^{N-1} (-2/n)(-1)^n sin(n a)  (note: N-1 terms, as range(1, N))."""
    y = np.zeros_like(a)
    for n in range(1, N):
        y = y + (-2.0 / n) * ((-1.0) ** n) * np.sin(n * a)
    return y


def _links(x, T, X):
    x = np.asarray(x)
    return x.reshape(x.shape[0], T, X, 2)


def plaq_sums(x, T, X):
    """gauge_model.py:659-681 / lattice.py:300-303.

    P[b,i,j] = x0[i,j] - x1[i,j] - x0[i,j+1] + x1[i+1,j]   (periodic);
    tf.roll(shift=-1, axis=2) brings j+1 to j, axis=1 brings i+1 to i.
    """
    s = _links(x, T, X)
    x0 = s[..., 0]
    x1 = s[..., 1]
    return x0 - x1 - np.roll(x0, -1, axis=2) + np.roll(x1, -1, axis=1)


def total_action(x, T, X):
    """lattice.py:337-362: S[b] = sum_ij (1 - cos P)."""
    return np.sum(1.0 - np.cos(plaq_sums(x, T, X)), axis=(1, 2))


def avg_plaq(x, T, X):
    """gauge_model.py:692-699: sum cos P / num_plaquettes (= T*X)."""
    return np.sum(np.cos(plaq_sums(x, T, X)), axis=(1, 2)) / (T * X)


def top_charge(x, T, X, fft=False):
    """gauge_model.py:701-716: sum project(P) / 2pi (real-valued, no rounding)."""
    p = plaq_sums(x, T, X)
    proj = project_angle_approx(p) if fft else project_angle(p)
    return np.sum(proj, axis=(1, 2)) / TWO_PI


def top_charge_floor(x, T, X):
    """lattice.py:309-311 variant: floor(0.1 + sum project(P)/2pi)."""
    return np.floor(0.1 + top_charge(x, T, X))


def top_charge_diff(x1, x2, T, X, fft=False):
    """gauge_model.py:718-725."""
    return np.abs(top_charge(x1, T, X, fft) - top_charge(x2, T, X, fft))


def calc_plaq_observables(x, T, X):
    """lattice.py:285-313 -> (total_action, avg_plaq, topological_charge)."""
    return total_action(x, T, X), avg_plaq(x, T, X), top_charge_floor(x, T, X)


def grad_action(x, T, X):
    """dS/dx, flat [B, D].  gauge_lattice.py:427-459 states it per link:
        dS/dx0[i,j] =  sin P[i,j] - sin P[i,j-1]
        dS/dx1[i,j] = -sin P[i,j] + sin P[i-1,j]
    (what tf.gradients of total_action yields, gauge_dynamics.py:698-709)."""
    sp = np.sin(plaq_sums(x, T, X))
    g = np.empty(sp.shape + (2,), dtype=sp.dtype)
    g[..., 0] = sp - np.roll(sp, 1, axis=2)
    g[..., 1] = -sp + np.roll(sp, 1, axis=1)
    return g.reshape(g.shape[0], -1)


def init_links(T, X, num_samples, rand, rng=None, dtype=np.float32):
    """lattice.py:95-162,208-241: cold start = zeros; hot start = U[0, 2pi)
    per link, sample 0 overwritten by the lattice's own `links` draw.  `rng`
    is a numpy RandomState (the reference uses the global legacy stream)."""
    shape = (T, X, 2)
    if not rand:
        return np.zeros((num_samples,) + shape, dtype=dtype).reshape(num_samples, -1)
    rng = rng or np.random
    links = np.array(rng.uniform(0, TWO_PI, shape), dtype=np.float32)
    samples = np.array([TWO_PI * rng.rand(*shape) for _ in range(num_samples)])
    samples[0] = links
    return samples.reshape(num_samples, -1).astype(dtype)
