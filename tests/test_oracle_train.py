"""CPU checks of the differentiable oracle (oracle/torch_ref.py) that the GPU gradient tests lean on:
it must agree with the NumPy restatement (which the reference's known answers pin, tests/test_oracle_kat.py),
its autograd gradients must agree with finite differences, and it must reproduce the committed fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import dynamics as od
from oracle import loss as oloss
from oracle.torch_ref import TorchDynamicsModel, TorchGaugeModel
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")
tt = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))   # noqa: E731


@pytest.mark.parametrize("arch,L", [("generic", 4), ("conv3D", 8)])
def test_differentiable_oracle_equals_numpy_oracle(arch, L):
    T = X = L
    N, eps, B, D = 2, 0.15, 5, 2 * L * L
    xp, vp = (H.gauge_weights if arch == 'generic' else H.conv_weights)(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp, arch=arch)
    tm = TorchGaugeModel(T, X, N, eps, orc.mask, xp, vp, arch=arch)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D)
    got = tm.apply_transition(tt(x), 2.0, tt(v0f), tt(v0b), tt(coin), tt(u))
    want = orc.apply_transition(x, 2.0, v0f, v0b, coin, u)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.detach().numpy(), w, atol=1e-12)
    rng = np.random.default_rng(0)
    z = rng.standard_normal((B, D))
    dz = (rng.standard_normal((B, D)), rng.standard_normal((B, D)), rng.uniform(size=B), rng.uniform(size=B))
    loss, _ = tm.loss(tt(x), tt(z), 2.0, tuple(map(tt, (v0f, v0b, coin, u))), tuple(map(tt, dz)))
    want_loss = oloss.calc_loss(orc, x, 2.0, (v0f, v0b, coin, u), z, dz)[0]
    assert float(loss.detach()) == pytest.approx(want_loss, rel=1e-12)


def test_differentiable_generic_dynamics_equals_numpy_oracle():
    tgt, N, eps, B = H.mog_target_oracle(), 4, 0.1, 7
    xp, vp = H.mlp_weights(2, 50, regime="stress")
    masks = od.make_masks(N, 2, np.random.RandomState(3))
    orc = od.DynamicsOracle(2, tgt, N, eps, masks, xp, vp)
    tm = TorchDynamicsModel(tgt, N, eps, masks, xp, vp)
    rng = np.random.default_rng(0)
    x = tgt.get_samples(B, rng)
    v0f, v0b, bits = rng.standard_normal((B, 2)), rng.standard_normal((B, 2)), rng.integers(0, 2, B)
    Lx, px = tm.propose(tt(x), tt(v0f), tt(v0b), tt(bits))
    want = od.propose(x, orc, v0f, v0b, bits)
    np.testing.assert_allclose(Lx.detach().numpy(), want[0], atol=1e-12)
    np.testing.assert_allclose(px.detach().numpy(), want[2], atol=1e-12)


def test_autograd_gradients_match_finite_differences():
    """d loss / d eps and d loss / d (one weight) of the lattice loss by central differences."""
    T = X = 4
    N, B, D = 2, 4, 32
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, N, 0.2, xp, vp)
    rng = np.random.default_rng(3)
    x, z = rng.uniform(0, 2 * np.pi, (B, D)), rng.standard_normal((B, D))
    mk = lambda: tuple(map(tt, (rng.standard_normal((B, D)), rng.standard_normal((B, D)), rng.uniform(size=B),   # noqa: E731
                                rng.uniform(size=B))))
    dx, dz = mk(), mk()

    def loss_at(eps, bump=0.0):
        w = {k: v.copy() for k, v in xp.items()}
        w['h_layer/W'][3, 5] += bump
        tm = TorchGaugeModel(T, X, N, eps, orc.mask, w, vp)
        return tm, tm.loss(tt(x), tt(z), 2.0, dx, dz)[0]
    tm, loss = loss_at(0.2)
    loss.backward()
    h = 1e-6
    fd_eps = (float(loss_at(0.2 + h)[1].detach()) - float(loss_at(0.2 - h)[1].detach())) / (2 * h)
    fd_w = (float(loss_at(0.2, h)[1].detach()) - float(loss_at(0.2, -h)[1].detach())) / (2 * h)
    assert float(tm.eps.grad) == pytest.approx(fd_eps, rel=1e-5)
    assert float(tm.xnet['h_layer/W'].grad[3, 5]) == pytest.approx(fd_w, rel=1e-4, abs=1e-9)


def _check_fixture_grads(g, prefix, grads):
    for k, v in grads.items():
        v = v.grad.numpy()
        if f"{prefix}/{k}" in g.files:
            np.testing.assert_allclose(v, g[f"{prefix}/{k}"], rtol=1e-9, atol=1e-12)
        else:
            np.testing.assert_allclose([v.sum(), np.abs(v).sum(), (v * v).sum()], g[f"{prefix}/{k}/stats"], rtol=1e-9)
            np.testing.assert_allclose(v.reshape(v.shape[0], -1)[0], g[f"{prefix}/{k}/row0"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["train_L8_generic", "train_L8_conv3d"])
def test_lattice_training_fixture_is_reproduced(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    T, X, N, arch = int(g["T"]), int(g["X"]), int(g["num_steps"]), str(g["arch"])
    xp, vp = (H.gauge_weights if arch == 'generic' else H.conv_weights)(T, X, seed=106, regime=str(g["regime"]))
    tm = TorchGaugeModel(T, X, N, float(g["eps"]), g["masks"], xp, vp, arch=arch)
    dx = tuple(tt(g[f"draws_x/{i}"]) for i in range(4))
    dz = tuple(tt(g[f"draws_z/{i}"]) for i in range(4))
    loss, terms = tm.loss(tt(g["x"]), tt(g["z"]), float(g["beta"]), dx, dz)
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-10)
    np.testing.assert_allclose(terms.detach().numpy(), g["terms"], rtol=1e-9)
    assert float(tm.eps.grad) == pytest.approx(float(g["grad_eps"]), rel=1e-9)
    _check_fixture_grads(g, "xnet", tm.xnet)
    _check_fixture_grads(g, "vnet", tm.vnet)


def test_mog_training_fixture_is_reproduced():
    g = np.load(os.path.join(GOLD, "train_mog.npz"))
    N = int(g["trajectory_length"])
    xp, vp = H.mlp_weights(2, int(g["num_nodes"]), seed=106, regime=str(g["regime"]))
    tm = TorchDynamicsModel(H.mog_target_oracle(), N, float(g["eps"]), g["masks"], xp, vp)
    dx = tuple(tt(g[f"draws_x/{i}"]) for i in range(4))
    dz = tuple(tt(g[f"draws_z/{i}"]) for i in range(4))
    loss, Lx, px, Lz, pz = tm.mog_loss(tt(g["x"]), tt(g["z"]), dx, dz, float(g["scale"]))
    loss.backward()
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-10)
    np.testing.assert_allclose(Lx.detach().numpy(), g["Lx"], rtol=1e-9, atol=1e-12)
    assert float(tm.alpha.grad) == pytest.approx(float(g["grad_alpha"]), rel=1e-9)
    _check_fixture_grads(g, "xnet", tm.xnet)
    _check_fixture_grads(g, "vnet", tm.vnet)
