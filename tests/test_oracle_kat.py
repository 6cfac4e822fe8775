"""Pins the CPU oracle against every known answer the reference holds for the
hot path: the reference has no tests or golden tensors, so these
are the exact-physics anchors, the frozen NumPy mask stream, the two statements
of the force, and the algebraic identities its docstrings claim."""
import numpy as np
import pytest
import torch

from oracle import lattice as lat
from oracle import nets
from oracle import dynamics as gen
from oracle.gauge_dynamics import GaugeDynamicsOracle, make_masks


def test_u1_plaq_exact_values():
    # lattice.py:31-33; values quoted in BASELINE.md / notebook header column "(EXACT)"
    assert abs(lat.u1_plaq_exact(2.0) - 0.697775) < 1e-6
    assert abs(lat.u1_plaq_exact(4.0) - 0.863523) < 1e-6


def test_cold_start_observables():
    # notebooks/gauge_model_graph_mode.ipynb:255-265 step-0 row: ACTION 0, PLAQ 1
    x = lat.init_links(8, 8, 5, rand=False)
    assert x.shape == (5, 128)
    S, plaq, Q = lat.calc_plaq_observables(x, 8, 8)
    assert np.all(S == 0) and np.all(plaq == 1) and np.all(Q == 0)
    assert np.all(lat.top_charge(x, 8, 8) == 0)


def test_mask_stream_seed_42():
    # globals.py:12 + gauge_model.py:195 seed the legacy global stream; first mask
    # for D=128 (gauge_dynamics.py:651-661) -- indices quoted in SURVEY.md 8a/a5.
    np.random.seed(42)
    m = make_masks(10, 128)
    assert m.shape == (10, 128)
    assert np.all(m.sum(axis=1) == 64)
    np.random.seed(42)
    first = np.random.permutation(np.arange(128))[:64]
    assert list(first[:8]) == [55, 40, 19, 31, 98, 56, 69, 104]
    assert np.all(m[0][first] == 1)


@pytest.mark.parametrize("T,X", [(8, 8), (4, 6), (16, 16)])
def test_force_analytic_equals_autodiff(T, X):
    # gauge_dynamics.py:698-709 (autodiff of the roll/cos action) vs
    # gauge_lattice.py:427-459 (hand-written) -- the reference's two statements.
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 2 * np.pi, (3, 2 * T * X))
    xt = torch.tensor(x, requires_grad=True)
    s = xt.reshape(3, T, X, 2)
    P = (s[..., 0] - s[..., 1] - torch.roll(s[..., 0], -1, 2) + torch.roll(s[..., 1], -1, 1))
    S = torch.sum(1. - torch.cos(P), dim=(1, 2))
    g, = torch.autograd.grad(S.sum(), xt)
    np.testing.assert_allclose(lat.total_action(x, T, X), S.detach().numpy(), rtol=0, atol=1e-12)
    np.testing.assert_allclose(lat.grad_action(x, T, X), g.numpy(), rtol=0, atol=1e-13)


def test_project_angle_range_and_charge_integer():
    rng = np.random.default_rng(1)
    a = rng.uniform(-4 * np.pi, 4 * np.pi, 1000)
    p = lat.project_angle(a)
    assert np.all(p >= -np.pi - 1e-12) and np.all(p < np.pi + 1e-12)
    np.testing.assert_allclose(np.exp(1j * p), np.exp(1j * a), atol=1e-12)
    x = rng.uniform(0, 2 * np.pi, (16, 128))
    q = lat.top_charge(x, 8, 8)
    np.testing.assert_allclose(q, np.round(q), atol=1e-9)   # sum of plaquettes is 0 mod 2pi


def _stress_oracle(T=4, X=4, num_steps=3, arch='generic', hmc=False, dtype=np.float64, seed=7):
    rng = np.random.default_rng(seed)
    D = 2 * T * X
    masks = make_masks(num_steps, D, np.random.RandomState(42))
    if arch == 'generic':
        xp = nets.init_generic_net(rng, D, 4 * D, 2., head_factor=0.1, bias_std=0.05, coeff_std=0.2)
        vp = nets.init_generic_net(rng, D, 4 * D, 1., head_factor=0.1, bias_std=0.05, coeff_std=0.2)
    else:
        xp = nets.init_conv3d_net(rng, T, D, 2 * D, T, 2., head_factor=0.1, bias_std=0.05, coeff_std=0.2)
        vp = nets.init_conv3d_net(rng, T, D, 2 * D, T, 1., head_factor=0.1, bias_std=0.05, coeff_std=0.2)
    return GaugeDynamicsOracle(T, X, num_steps, 0.2, masks, xp, vp, arch, hmc=hmc, dtype=dtype), rng


@pytest.mark.parametrize("arch", ['generic', 'conv3D'])
def test_backward_lf_inverts_forward_lf(arch):
    # gauge_dynamics.py:537-590 docstrings: "_update_*_backward ... Invert the forward update"
    dyn, rng = _stress_oracle(arch=arch)
    x = rng.uniform(0, 2 * np.pi, (5, dyn.x_dim))
    v = rng.standard_normal((5, dyn.x_dim))
    for step in range(dyn.num_steps):
        x1, v1, ld_f = dyn._forward_lf(x, v, 2.5, step)
        # _backward_lf reverses the index internally (:453-457)
        x2, v2, ld_b = dyn._backward_lf(x1, v1, 2.5, dyn.num_steps - 1 - step)
        np.testing.assert_allclose(x2, x, atol=1e-10)
        np.testing.assert_allclose(v2, v, atol=1e-10)
        np.testing.assert_allclose(ld_b, -ld_f, atol=1e-10)
        assert np.max(np.abs(ld_f)) > 1e-3      # the stress regime really exercises the log-det


def test_sumlogdet_is_log_abs_det_jacobian():
    dyn, rng = _stress_oracle(T=2, X=2, num_steps=2)
    D = dyn.x_dim
    x = rng.uniform(0, 2 * np.pi, (1, D))
    v = rng.standard_normal((1, D))

    def flow(z):
        a, b, _ = dyn._forward_lf(z[None, :D], z[None, D:], 2.0, 1)
        return np.concatenate([a[0], b[0]])

    z0 = np.concatenate([x[0], v[0]])
    h = 1e-6
    J = np.stack([(flow(z0 + h * e) - flow(z0 - h * e)) / (2 * h) for e in np.eye(2 * D)], axis=1)
    _, _, ld = dyn._forward_lf(x, v, 2.0, 1)
    sign, logabs = np.linalg.slogdet(J)
    assert abs(logabs - ld[0]) < 1e-6


def test_hmc_mode_is_plain_leapfrog():
    # gauge_dynamics.py:102-108: S=T=Q=0 => v half-kick, x += eps v on both mask halves, half-kick
    dyn, rng = _stress_oracle(T=4, X=4, num_steps=4, hmc=True)
    x = rng.uniform(0, 2 * np.pi, (6, dyn.x_dim))
    v = rng.standard_normal((6, dyn.x_dim))
    beta, eps = 2.0, dyn.eps
    x1, v1, ld = dyn._forward_lf(x, v, beta, 0)
    vh = v - 0.5 * eps * dyn.grad_potential(x, beta)
    xr = x + eps * vh
    vr = vh - 0.5 * eps * dyn.grad_potential(xr, beta)
    np.testing.assert_allclose(x1, xr, atol=1e-12)
    np.testing.assert_allclose(v1, vr, atol=1e-12)
    assert np.all(ld == 0)
    # symplectic + time-reversible: small energy error, p in (0, 1]
    errs = []
    for e in (0.04, 0.02):
        dyn.eps = np.float64(e)
        xN, vN, p, sld = dyn.transition_kernel(x, beta, v, forward=True)
        dH = dyn.hamiltonian(xN, vN, beta) - dyn.hamiltonian(x, v, beta)
        assert np.all(sld == 0) and np.all((p > 0) & (p <= 1))
        errs.append(np.max(np.abs(dH)))
    assert errs[0] < 0.1 and errs[1] < 0.4 * errs[0]      # O(eps^2) energy error


def test_apply_transition_mixing_and_strict_accept():
    dyn, rng = _stress_oracle()
    B, D = 8, dyn.x_dim
    x = rng.uniform(0, 2 * np.pi, (B, D))
    vf, vb = rng.standard_normal((B, D)), rng.standard_normal((B, D))
    coin = np.array([0.1, 0.9, 0.5, 0.51, 0.3, 0.7, 0.2, 0.8])
    u = rng.uniform(size=B)
    xp, vp, p, xo = dyn.apply_transition(x, 2.0, vf, vb, coin, u)
    xf, vf_, pf, _ = dyn.transition_kernel(x, 2.0, vf, True)
    xb, vb_, pb, _ = dyn.transition_kernel(x, 2.0, vb, False)
    for i in range(B):
        fwd = coin[i] > 0.5                       # :221-227 (0.5 itself -> backward)
        np.testing.assert_array_equal(xp[i], xf[i] if fwd else xb[i])
        np.testing.assert_array_equal(vp[i], vf_[i] if fwd else vb_[i])
        assert p[i] == (pf[i] if fwd else pb[i])
        np.testing.assert_array_equal(xo[i], xp[i] if p[i] > u[i] else x[i])
    # strict '>' (Q5): u == p rejects
    _, _, p2, xo2 = dyn.apply_transition(x, 2.0, vf, vb, coin, p)
    np.testing.assert_array_equal(xo2, x)


def test_fp32_mode_tracks_fp64():
    d64, _ = _stress_oracle(T=8, X=8, num_steps=5, dtype=np.float64)
    d32, rng = _stress_oracle(T=8, X=8, num_steps=5, dtype=np.float32)
    x = rng.uniform(0, 2 * np.pi, (4, 128))
    v = rng.standard_normal((4, 128))
    a = d64.transition_kernel(x, 2.0, v, True)
    b = d32.transition_kernel(x.astype(np.float32), 2.0, v.astype(np.float32), True)
    assert b[0].dtype == np.float32
    assert np.max(np.abs(a[0] - b[0])) / np.max(np.abs(a[0])) < 1e-4


def test_conv3d_front_matches_torch_ops():
    """Independent check of the restated Keras 'same' conv / pool conventions
    (conv_net.py:90-164) against torch's conv3d / max_pool3d with the padding
    written out by hand."""
    import torch.nn.functional as Fn
    rng = np.random.default_rng(3)
    L, F = 8, 8
    p = nets.init_conv3d_net(rng, L, 2 * L * L, 4 * L * L, F, 2., bias_std=0.1)
    a = rng.standard_normal((3, 2 * L * L))
    got = nets.conv3d_front(p, a, 'x', (L, L, 2))

    t = torch.tensor(a).reshape(3, L, L, 2, 1).permute(0, 4, 1, 2, 3)       # N C D H W
    w1 = torch.tensor(p['conv_x1/W']).permute(4, 3, 0, 1, 2)
    t = Fn.pad(t, (0, 1, 1, 1, 1, 1))        # last axis (k=2): 0 before / 1 after; k=3 axes: 1/1
    t = torch.relu(Fn.conv3d(t, w1, torch.tensor(p['conv_x1/b'])))
    t = Fn.max_pool3d(t, 2, 2)               # 8,8,2 -> 4,4,1 (no padding needed)
    w2 = torch.tensor(p['conv_x2/W']).permute(4, 3, 0, 1, 2)
    t = Fn.pad(t, (0, 1, 0, 1, 0, 1))
    t = torch.relu(Fn.conv3d(t, w2, torch.tensor(p['conv_x2/b'])))
    t = Fn.max_pool3d(t, (2, 2, 1), (2, 2, 1))     # depth axis already 1: 'same' keeps it
    want = t.permute(0, 2, 3, 4, 1).reshape(3, -1).numpy()
    assert got.shape == (3, nets.conv3d_flat_size(L, F)) == (3, 64)
    np.testing.assert_allclose(got, want, atol=1e-12)


def _torch_energy_grad(fn, x):
    xt = torch.tensor(x, requires_grad=True)
    e = fn(xt)
    g, = torch.autograd.grad(e.sum(), xt)
    return e.detach().numpy(), g.numpy()


def test_gmm_and_gaussian_energy_gradients():
    # utils/dynamics.py:241-242 differentiates distributions.py:151-158 / :63-68
    rng = np.random.default_rng(5)
    x = rng.standard_normal((32, 2))
    g = gen.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])

    def gmm_t(xt):
        cols = []
        for i in range(2):
            d = xt - torch.tensor(g.mus[i].astype('float32').astype('float64'))
            S = torch.tensor(g.i_sigmas[i].astype('float64'))
            cols.append(-0.5 * torch.einsum('bi,ij,bj->b', d, S, d) + float(np.log(g.constants[i])))
        return -torch.logsumexp(torch.stack(cols, 1), dim=1)

    e, gr = _torch_energy_grad(gmm_t, x)
    np.testing.assert_allclose(g.energy(x), e, atol=1e-6)
    np.testing.assert_allclose(g.grad_energy(x), gr, atol=1e-5)

    cov = np.array([[50.05, -49.95], [-49.95, 50.05]])
    ga = gen.Gaussian(np.zeros(2), cov)
    S = torch.tensor(ga.i_sigma.astype('float32').astype('float64'))

    def ga_t(xt):
        # the reference's BxB form (distributions.py:36-39)
        return torch.diagonal(0.5 * (xt @ S) @ xt.T)

    e, gr = _torch_energy_grad(ga_t, x)
    np.testing.assert_allclose(ga.energy(x), e, atol=1e-10)
    np.testing.assert_allclose(ga.grad_energy(x), gr, atol=1e-10)


def test_generic_dynamics_inverse_and_propose():
    rng = np.random.default_rng(9)
    xp = nets.init_mlp_net(rng, 2, 2., 10, head_factor=0.1, bias_std=0.05, coeff_std=0.2)
    vp = nets.init_mlp_net(rng, 2, 1., 10, head_factor=0.1, bias_std=0.05, coeff_std=0.2)
    masks = gen.make_masks(5, 2, np.random.RandomState(42))
    tgt = gen.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    d = gen.DynamicsOracle(2, tgt, 5, 0.1, masks, xp, vp)
    x = tgt.get_samples(16, rng)
    v = rng.standard_normal((16, 2))
    for step in range(5):
        x1, v1, lf = d._forward_step(x, v, step)
        x2, v2, lb = d._backward_step(x1, v1, step)
        np.testing.assert_allclose(x2, x, atol=1e-10)
        np.testing.assert_allclose(v2, v, atol=1e-10)
        np.testing.assert_allclose(lb, -lf, atol=1e-10)
    bits = rng.integers(0, 2, 16)
    u = rng.uniform(size=16)
    Lx, Lv, px, outs, Lvm = gen.propose(x, d, v, v[::-1].copy(), bits, u, do_mh_step=True)
    assert Lv is None and len(outs) == 1          # sampler.py:43-45
    X1, _, p1 = d.forward(x, v)
    X2, _, p2 = d.backward(x, v[::-1].copy())
    for i in range(16):
        np.testing.assert_array_equal(Lx[i], X1[i] if bits[i] else X2[i])
        acc = px[i] - u[i] >= 0                   # sampler.py:58 non-strict
        np.testing.assert_array_equal(outs[0][i], Lx[i] if acc else x[i])
