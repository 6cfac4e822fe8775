"""Shared builders for the parity tests: an oracle and a HIP-backed dynamics
object holding the same weights, masks and step size."""
import numpy as np

from oracle import nets
from oracle import dynamics as ogen
from oracle.gauge_dynamics import GaugeDynamicsOracle, make_masks

REGIMES = {
    # reference initialisation (generic_net.py:39-90): heads factor 0.001, zero biases/coeffs
    "init": dict(head_factor=0.001, bias_std=0.0, coeff_std=0.0),
    # SURVEY.md 8d "stress": eps*S, eps*Q reach O(0.3) so exp / tanh / log-det paths matter
    "stress": dict(head_factor=0.1, bias_std=0.05, coeff_std=0.2),
    # same paths exercised, gentle enough to stay finite over 10+ leapfrog steps at D=128
    "mild": dict(head_factor=0.01, bias_std=0.02, coeff_std=0.1),
}


def relerr(got, want):
    """max |got - want| relative to the tensor's own scale (>= 1)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    return float(np.max(np.abs(got - want)) / max(1.0, np.max(np.abs(want)))) if want.size else 0.0


def gauge_weights(T, X, seed=106, regime="stress", hidden_mult=4):
    rng = np.random.default_rng(seed)
    D = 2 * T * X
    kw = REGIMES[regime]
    xp = nets.init_generic_net(rng, D, hidden_mult * D, 2., **kw)
    vp = nets.init_generic_net(rng, D, hidden_mult * D, 1., **kw)
    return xp, vp


def conv_weights(T, X, seed=106, regime="stress"):
    """ConvNet3D as gauge_dynamics.py:121-143 builds it: F = space_size, H = 2 * x_dim."""
    rng = np.random.default_rng(seed)
    D = 2 * T * X
    kw = REGIMES[regime]
    xp = nets.init_conv3d_net(rng, T, D, 2 * D, X, 2., **kw)
    vp = nets.init_conv3d_net(rng, T, D, 2 * D, X, 1., **kw)
    return xp, vp


def gauge_oracle(T, X, num_steps, eps, xp, vp, hmc=False, dtype=np.float64, mask_seed=42, arch='generic'):
    masks = make_masks(num_steps, 2 * T * X, np.random.RandomState(mask_seed))
    return GaugeDynamicsOracle(T, X, num_steps, eps, masks, xp, vp, arch, hmc=hmc, dtype=dtype)


def gauge_hip(T, X, num_steps, eps, xp, vp, masks, batch, hmc=False, both_directions=True, arch='generic'):
    from l2hmc_amd import GaugeLattice, GaugeDynamics
    lat = GaugeLattice(T, X, 2, 'U1', num_samples=batch, rand=False)
    dyn = GaugeDynamics(lat, lat.get_energy_function(), eps=eps, hmc=hmc, network_arch=arch,
                        num_steps=num_steps, eps_trainable=True, data_format='channels_last',
                        both_directions=both_directions)
    dyn.set_masks(masks)
    if not hmc:
        dyn.position_fn.load_state(xp)
        dyn.momentum_fn.load_state(vp)
    return dyn


def gauge_inputs(B, D, seed=103):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 2 * np.pi, (B, D))
    v0f = np.random.default_rng(seed + 1000).standard_normal((B, D))
    v0b = np.random.default_rng(seed + 1001).standard_normal((B, D))
    coin = np.random.default_rng(seed + 2000).uniform(size=B)
    u = np.random.default_rng(seed + 3000).uniform(size=B)
    return x, v0f, v0b, coin, u


def mog_target_oracle():
    """cfg 2: means on the axes, sigma^2 = 0.025 I, equal weights (mog_model.py:1063-1120)."""
    return ogen.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])


def scg_target_oracle():
    """cfg 1: strongly correlated Gaussian (SCGExperiment.ipynb cell 3)."""
    return ogen.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))


def mlp_weights(x_dim, num_nodes, seed=106, regime="stress"):
    rng = np.random.default_rng(seed)
    kw = REGIMES[regime]
    return (nets.init_mlp_net(rng, x_dim, 2., num_nodes, **kw),
            nets.init_mlp_net(rng, x_dim, 1., num_nodes, **kw))
