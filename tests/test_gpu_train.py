"""GPU parity of the training path: gradients of the loss with respect to every weight and
the step size, through the C ABI, against torch.autograd on the float64 restatement (oracle/torch_ref.py).

Tolerance: the forward path meets north_star's 1e-5; gradients are sums of ~1e5 fp32 products through up to
4N chained network calls, so they are compared per tensor in the max norm relative to the tensor's own largest
entry at 2e-4 (measured: 1e-6..3e-5)."""
import os

import numpy as np
import pytest
import torch

from oracle.torch_ref import TorchGaugeModel
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_G = 2e-4


def _setup(L, N, eps, B, regime, metric='cos_diff', seed=7, arch='generic'):
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    T = X = L
    D = 2 * T * X
    xp, vp = H.gauge_weights(T, X, regime=regime) if arch == 'generic' else H.conv_weights(T, X, regime=regime)
    orc = H.gauge_oracle(T, X, N, eps, xp, vp, arch=arch)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B, arch=arch)
    tr = GaugeTrainer(dyn, metric=metric, lr_init=1e-3)
    tm = TorchGaugeModel(T, X, N, eps, orc.mask, xp, vp, arch=arch)
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 2 * np.pi, (B, D))
    z = rng.standard_normal((B, D))
    dx = (rng.standard_normal((B, D)), rng.standard_normal((B, D)), rng.uniform(size=B), rng.uniform(size=B))
    dz = (rng.standard_normal((B, D)), rng.standard_normal((B, D)), rng.uniform(size=B), rng.uniform(size=B))
    return tr, tm, x, z, dx, dz


def _ref_grads(tm, x, z, dx, dz, beta, metric, **w):
    tt = lambda a: torch.tensor(a, dtype=torch.float64)
    loss, terms = tm.loss(tt(x), tt(z), beta, tuple(map(tt, dx)), tuple(map(tt, dz)), metric=metric, **w)
    loss.backward()
    return float(loss.detach()), terms.detach().numpy()


def _packed_ref(net):
    """autograd gradients of one network, rearranged into the library's k-contiguous layout."""
    g = {k: v.grad.numpy() for k, v in net.items()}
    conv = {}
    if 'conv_v1/W' in g:       # ConvNet3D: Keras-layout kernels; *_a = first input (conv_v*), *_b = second (conv_x*)
        conv = {"w1_a": g['conv_v1/W'], "b1_a": g['conv_v1/b'], "w2_a": g['conv_v2/W'], "b2_a": g['conv_v2/b'],
                "w1_b": g['conv_x1/W'], "b1_b": g['conv_x1/b'], "w2_b": g['conv_x2/W'], "b2_b": g['conv_x2/b']}
    return {
        **conv,
        "w1_t": np.concatenate([g['v_layer/W'], g['x_layer/W']], axis=0).T,
        "wt": g['t_layer/W'],
        "b1": g['v_layer/b'],
        "wh_t": g['h_layer/W'].T,
        "bh": g['h_layer/b'],
        "whd_t": np.stack([g['scale_layer/W'].T, g['translation_layer/W'].T, g['transformation_layer/W'].T]),
        "bhd": np.stack([g['scale_layer/b'], g['translation_layer/b'], g['transformation_layer/b']]),
        "coeff_s": g['coeff_scale'].reshape(-1),
        "coeff_q": g['coeff_transformation'].reshape(-1),
    }


def _compare(tr, tm, tol=TOL_G):
    gv = tr.grad_views()
    worst = {}
    for name, net in (("xnet", tm.xnet), ("vnet", tm.vnet)):
        ref = _packed_ref(net)
        # the three first-layer biases have identical gradients in the reference graph
        assert np.allclose(net['v_layer/b'].grad.numpy(), net['x_layer/b'].grad.numpy())
        assert np.allclose(net['v_layer/b'].grad.numpy(), net['t_layer/b'].grad.numpy())
        for k, want in ref.items():
            got = gv[name][k].cpu().numpy().astype(np.float64).reshape(want.shape)
            scale = np.abs(want).max()
            assert scale > 0, (name, k)
            worst[f"{name}.{k}"] = float(np.abs(got - want).max() / scale)
    want = float(tm.eps.grad)
    worst["eps"] = abs(float(gv["eps"][0]) - want) / abs(want)
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, f"gradient mismatch: {bad}\nall: {worst}"
    return worst


@pytest.mark.parametrize("L,N,eps,B,regime,fused", [
    (4, 3, 0.2, 6, "mild", True),
    (4, 2, 0.15, 37, "stress", True),       # ragged batch, strong S/Q
    (8, 2, 0.1, 16, "mild", True),          # benchmark widths D=128, H=512: whole-trajectory forward + reverse kernels
    (8, 3, 0.1, 9, "mild", True),           # ... with a partly filled 16-row tile (18 rows)
    (8, 2, 0.1, 16, "mild", False),         # same widths through the layered kernels
    (16, 1, 0.1, 3, "mild", True),          # D=512, H=2048: multi-group column sums, 16 x 16 output tiles of the TN products
    (32, 1, 0.1, 2, "mild", True),          # cfg 5's width D=2048, H=8192 (gauge_dynamics.py:169-187): 151 M weights per net
])
def test_loss_gradients_match_autograd(L, N, eps, B, regime, fused):
    tr, tm, x, z, dx, dz = _setup(L, N, eps, B, regime)
    tr.dynamics.fused = fused
    beta = 2.5
    loss, x_out, px, x_dq = tr.calc_loss_and_grads(x, beta, z=z, draws_x=dx, draws_z=dz)
    want_loss, want_terms = _ref_grads(tm, x, z, dx, dz, beta, 'cos_diff')
    # per-chain terms are differences of O(1/eps_loss = 1e3) quantities: absolute tolerance on that scale
    np.testing.assert_allclose(tr.last_loss_terms.cpu().numpy(), want_terms, rtol=2e-4,
                               atol=1e-5 * max(1., np.abs(want_terms).max()))
    assert abs(float(loss) - want_loss) <= 2e-4 * max(1., abs(want_loss))
    _compare(tr, tm)


@pytest.mark.parametrize("metric", ['l1', 'l2', 'cos', 'cos2'])
def test_loss_gradients_other_metrics_and_weights(metric):
    tr, tm, x, z, dx, dz = _setup(4, 2, 0.2, 9, "mild", metric=metric)
    tr.loss_scale = 0.7
    tr.weights = dict(aux_weight=0.5, std_weight=1.3, charge_weight=0.8)
    tr.calc_loss_and_grads(x, 3.0, z=z, draws_x=dx, draws_z=dz)
    _ref_grads(tm, x, z, dx, dz, 3.0, metric, loss_scale=0.7, aux_weight=0.5, std_weight=1.3, charge_weight=0.8)
    _compare(tr, tm)


def test_taped_forward_equals_sampling_path():
    """The taped forward pass must produce what the sampling kernels produce (same C-ABI outputs)."""
    import ctypes as C
    from l2hmc_amd import _lib
    tr, tm, x, z, dx, dz = _setup(8, 3, 0.2, 12, "mild")
    dyn = tr.dynamics
    dev = dyn._device
    R, D = 24, 128
    x0 = _lib.as_dev(np.concatenate([x, z]), dev)
    v0 = _lib.as_dev(np.concatenate([dx[0], dz[0]]), dev)
    dirs = torch.tensor([0, 1] * 12, dtype=torch.int32, device=dev)
    plan, L = dyn._plan(), _lib.lib()
    outs = []
    for mode in ("train", "sample"):
        xo, vo = torch.empty_like(x0), torch.empty_like(x0)
        sld, p = torch.empty(R, device=dev), torch.empty(R, device=dev)
        if mode == "train":
            ws, nb = tr._ws.get(L.l2hmc_gauge_train_ws_bytes(C.byref(plan), R), dev)
            _lib.check(L.l2hmc_gauge_train_forward(C.byref(plan), 2.0, x0.data_ptr(), v0.data_ptr(), dirs.data_ptr(), R,
                                                   xo.data_ptr(), vo.data_ptr(), sld.data_ptr(), p.data_ptr(), ws, nb,
                                                   _lib.stream_ptr()))
        else:
            ws, nb = dyn._ws.get(L.l2hmc_gauge_ws_bytes(C.byref(plan), R), dev)
            _lib.check(L.l2hmc_gauge_trajectory(C.byref(plan), 2.0, x0.data_ptr(), v0.data_ptr(), dirs.data_ptr(), R,
                                                xo.data_ptr(), vo.data_ptr(), sld.data_ptr(), p.data_ptr(), ws, nb,
                                                _lib.stream_ptr()))
        outs.append([t.cpu().numpy() for t in (xo, vo, sld, p)])
    for a, b in zip(*outs):
        assert H.relerr(a, b) <= 2e-5


def test_adam_matches_reference_update_rule():
    """tf.train.AdamOptimizer's update (epsilon-hat form) on the flat buffers, with clip_by_global_norm, for 3
    steps of constant synthetic gradients; the packed first-layer bias moves three times as far."""
    tr, tm, x, z, dx, dz = _setup(4, 2, 0.2, 4, "mild")
    tr.clip_value = 0.5
    dyn = tr.dynamics
    flats = [n.flat_params() for n in tr._nets]
    w0 = [f[0].clone() for f in flats]
    eps0 = float(dyn.eps)
    g = torch.randn_like(tr.grads) * 0.01
    # numpy restatement
    n0, n1 = tr._sizes
    gn = g.cpu().numpy().astype(np.float64)
    tri = np.ones_like(gn)
    off = 0
    for f in flats:
        a, b = f[2]["b1"]
        tri[off + a:off + b] = 3.
        off += f[0].numel()
    norm = np.sqrt(np.sum(tri * gn * gn))
    gc = gn * 0.5 / max(norm, 0.5)
    m = np.zeros_like(gn); v = np.zeros_like(gn)
    w = np.concatenate([w0[0].cpu().numpy(), w0[1].cpu().numpy(), [eps0]]).astype(np.float64)
    for t in range(1, 4):
        tr.grads.copy_(g)
        lr = tr.learning_rate()
        tr.apply_gradients()
        m = 0.9 * m + 0.1 * gc
        v = 0.999 * v + 0.001 * gc * gc
        lr_t = lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        w = w - tri * lr_t * m / (np.sqrt(v) + 1e-8)
    got = np.concatenate([flats[0][0].cpu().numpy(), flats[1][0].cpu().numpy(), [float(dyn.eps)]])
    np.testing.assert_allclose(got, w, rtol=2e-5, atol=2e-7)
    assert tr.global_step == 3


def test_training_reduces_loss_and_sampler_sees_new_weights():
    """A few train steps on a fixed batch lower the loss; the fused sampling kernel then runs with the updated
    weights (its packed image is rebuilt) and agrees with the layered path."""
    tr, tm, x, z, dx, dz = _setup(8, 3, 0.15, 64, "init")
    tr.lr_init = 1e-4      # measured: monotone decrease -122.5 -> -145.4 over 8 steps on this batch
    losses = []
    for _ in range(6):
        loss, *_ = tr.train_step(x, 2.0, z=z, draws_x=dx, draws_z=dz)
        losses.append(float(loss))
    assert np.isfinite(losses).all()
    assert losses[1] < losses[0] and losses[-1] < losses[1], losses
    dyn = tr.dynamics
    xin, v0f, v0b, coin, u = H.gauge_inputs(64, 128)
    dyn.fused = True
    a = dyn.apply_transition(xin, 2.0, v0f, v0b, coin, u)
    dyn.fused = False
    b = dyn.apply_transition(xin, 2.0, v0f, v0b, coin, u)
    for s, t in zip(a, b):
        assert H.relerr(s.cpu().numpy(), t.cpu().numpy()) <= 5e-5
    # reference-layout tensors follow the flat buffer
    before = dyn.position_fn.h_layer.kernel.clone()
    tr.sync_weights()
    assert not torch.equal(before, dyn.position_fn.h_layer.kernel)
    np.testing.assert_array_equal(dyn.position_fn.h_layer.kernel.t().cpu().numpy(),
                                  dyn.position_fn.flat_params()[1]["wh_t"].cpu().numpy())


def test_data_parallel_gradients_equal_full_batch(tmp_path):
    """two ranks, each with half of the chains, one all-reduce -> the full-batch gradient and
    loss.  (Both ranks share the test box's single GPU; the exchange runs over gloo.  On a multi-GPU node the
    same code path runs with backend "nccl" = RCCL.)"""
    import os
    import socket
    import subprocess
    import sys
    B = 10
    tr, tm, x, z, dx, dz = _setup(4, 2, 0.2, B, "mild")
    loss, *_ = tr.calc_loss_and_grads(x, 2.5, z=z, draws_x=dx, draws_z=dz)
    full = tr.grads.cpu().numpy().copy()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "dp.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "dp_train_worker.py"),
                                       out, str(B)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    with np.load(out) as f:
        got, got_loss, lr = f["grads"], float(f["loss"]), float(f["lr"])
        # f2: the bucketed exchange (groups sent while later groups are still being computed) gives bit for bit
        # what a single all-reduce of the whole buffer gives
        np.testing.assert_array_equal(f["grads"], f["grads_single"])
    assert abs(got_loss - float(loss)) <= 1e-5 * max(1., abs(float(loss)))
    scale = np.abs(full).max()
    assert np.abs(got - full).max() <= 2e-5 * scale      # summation order differs between 1 and 2 shards
    assert lr == pytest.approx(2 * tr.learning_rate())    # gauge_model.py:942: lr * hvd.size()


def test_save_and_resume_reproduces_the_next_step(tmp_path):
    """gauge_model.py:519-556: a run restored from its saved state continues identically (weights, Adam moments,
    step size, counters, RNG draw counter)."""
    tr, tm, x, z, dx, dz = _setup(4, 2, 0.2, 8, "mild")
    for _ in range(3):
        tr.train_step(x, 2.0)                     # library-drawn z / momenta: exercises the draw counter
    path = str(tmp_path / "state.npz")
    tr.save_state(path, samples=x, beta=2.0)
    a = tr.train_step(x, 2.0)
    wa = [n.flat_params()[0].clone() for n in tr._nets]
    tr2, *_ = _setup(4, 2, 0.2, 8, "init", seed=99)       # different weights, masks of the same seed
    extra = tr2.load_state(path)
    assert float(extra["beta"]) == 2.0 and extra["samples"].shape == x.shape
    b = tr2.train_step(x, 2.0)
    assert float(a[0]) == float(b[0])
    for p, q in zip(wa, [n.flat_params()[0] for n in tr2._nets]):
        assert torch.equal(p, q)
    assert float(tr.dynamics.eps) == float(tr2.dynamics.eps) and tr.global_step == tr2.global_step


def test_train_loop_anneals_beta_and_keeps_samples_wrapped():
    tr, tm, x, z, dx, dz = _setup(4, 2, 0.1, 16, "init")
    tr.lr_init = 1e-4
    out = tr.train(5, samples_init=x, beta_init=2., beta_final=3.)
    assert out["loss"].shape == (5,) and out["charges"].shape == (5, 16) and np.isfinite(out["loss"]).all()
    want_beta = [1. / ((1. / 2 - 1. / 3) * (1 - s / 5.) + 1. / 3) for s in range(5)]       # gauge_model.py:1039-1046
    np.testing.assert_allclose(out["beta"], want_beta, rtol=1e-12)
    xs = out["samples"].cpu().numpy()
    assert (xs >= 0).all() and (xs < 2 * np.pi + 1e-6).all()
    assert tr.global_step == 5 and len(set(out["eps"])) > 1                                  # eps is being trained


@pytest.mark.parametrize("L,N,eps,B,regime", [(8, 2, 0.1, 9, "mild"), (8, 3, 0.15, 5, "stress"),
                                              (16, 1, 0.1, 3, "mild")])      # cfg 4's F = 16, 16x16 kernel instances
def test_conv3d_loss_gradients_match_autograd(L, N, eps, B, regime):
    """ConvNet3D (the reference's CLI-default architecture, conv_net.py:247-280) at the 8x8 benchmark lattice and at
    cfg 4's 16x16: gradients of the Conv3D kernels / biases (through both max-pools and relus), the dense trunk and
    eps."""
    tr, tm, x, z, dx, dz = _setup(L, N, eps, B, regime, arch='conv3D')
    loss, *_ = tr.calc_loss_and_grads(x, 2.5, z=z, draws_x=dx, draws_z=dz)
    want_loss, _ = _ref_grads(tm, x, z, dx, dz, 2.5, 'cos_diff')
    assert abs(float(loss) - want_loss) <= 2e-4 * max(1., abs(want_loss))
    worst = _compare(tr, tm)
    # the dd = 1 slice of the second kernel only multiplies padding
    gv = tr.grad_views()
    assert float(gv["xnet"]["w2_a"].reshape(2, 2, 2, L, 2 * L)[:, :, 1].abs().max()) == 0.0
    assert "xnet.w1_a" in worst and "vnet.b2_b" in worst


def test_conv3d_training_step_updates_filters_and_sampler():
    tr, tm, x, z, dx, dz = _setup(8, 2, 0.1, 32, "init", arch='conv3D')
    tr.lr_init = 1e-4
    dyn = tr.dynamics
    k0 = dyn.position_fn.flat_params()[1]["w1_a"].clone()
    losses = [float(tr.train_step(x, 2.0, z=z, draws_x=dx, draws_z=dz)[0]) for _ in range(4)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert not torch.equal(k0, dyn.position_fn.flat_params()[1]["w1_a"])
    xin, v0f, v0b, coin, u = H.gauge_inputs(32, 128)
    dyn.fused = True
    a = dyn.apply_transition(xin, 2.0, v0f, v0b, coin, u)
    dyn.fused = False
    b = dyn.apply_transition(xin, 2.0, v0f, v0b, coin, u)
    for s_, t_ in zip(a, b):
        assert H.relerr(s_.cpu().numpy(), t_.cpu().numpy()) <= 5e-5
    tr.sync_weights()
    np.testing.assert_array_equal(dyn.position_fn.conv_v1.kernel.cpu().numpy().reshape(-1),
                                  dyn.position_fn.flat_params()[1]["w1_a"].cpu().numpy().reshape(-1))


# ------------------------------------------------------------------ toy targets (mog_model.py:324-363)
def _small_setup(kind, H_nodes, N, eps, B, regime, seed=11, temperature=1.0):
    import l2hmc_amd as la
    from l2hmc_amd.dynamics_trainer import DynamicsTrainer
    from oracle import dynamics as od
    from oracle.torch_ref import TorchDynamicsModel
    dim = 2
    if kind == "mog":
        tgt_o = H.mog_target_oracle()
        tgt = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    elif kind == "mog3":      # x_dim 3: the run-time-dimension instances of the kernels (x_dim 2 has its own)
        dim = 3
        mus = [np.array([1., 0., 0.5]), np.array([0., 1., -0.5]), np.array([-1., -1., 0.])]
        covs = [np.diag([0.05, 0.08, 0.1]), 0.07 * np.eye(3) + 0.02, np.diag([0.1, 0.05, 0.06])]
        pis = [0.3, 0.5, 0.2]
        from oracle import dynamics as _od
        tgt_o, tgt = _od.GMM(mus, covs, pis), la.GMM(mus, covs, pis)
    else:
        tgt_o = H.scg_target_oracle()
        tgt = la.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
    xp, vp = H.mlp_weights(dim, H_nodes, regime=regime)
    masks = od.make_masks(N, dim, np.random.RandomState(3))
    dyn = la.Dynamics(dim, tgt.get_energy_function(), trajectory_length=N, eps=eps,
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=H_nodes),
                      use_temperature=True)
    dyn.temperature = temperature
    dyn.set_masks(masks)
    dyn.XNet.load_state(xp)
    dyn.VNet.load_state(vp)
    tr = DynamicsTrainer(dyn, scale=0.1)
    tm = TorchDynamicsModel(tgt_o, N, eps, masks, xp, vp, temperature=temperature)
    rng = np.random.default_rng(seed)
    x = tgt_o.get_samples(B, rng)
    z = rng.standard_normal((B, dim))
    mk = lambda: (rng.standard_normal((B, dim)), rng.standard_normal((B, dim)),   # noqa: E731
                  rng.integers(0, 2, B).astype(np.float64), rng.uniform(size=B))
    return tr, tm, x, z, mk(), mk()


def _mlp_packed_ref(net):
    g = {k: v.grad.numpy() for k, v in net.items()}
    return {
        "w1_t": np.concatenate([g['embed_1/W'], g['embed_2/W']], axis=0).T, "wt": g['embed_3/W'], "b1": g['embed_1/b'],
        "wh_t": g['linear_1/W'].T, "bh": g['linear_1/b'],
        "whd_t": np.stack([g['linear_s/W'].T, g['linear_t/W'].T, g['linear_f/W'].T]),
        "bhd": np.stack([g['linear_s/b'], g['linear_t/b'], g['linear_f/b']]),
        "coeff_s": g['scale_s'].reshape(-1), "coeff_q": g['scale_f'].reshape(-1),
    }


@pytest.mark.parametrize("kind,H_nodes,N,eps,B,regime,temp", [
    ("mog", 50, 5, 0.1, 37, "stress", 1.0),        # cfg 2 widths, ragged batch (3 workgroups, one partly empty)
    ("mog", 50, 10, 0.1, 16, "mild", 2.5),         # full trajectory length, tempered target
    ("scg", 10, 5, 0.1, 21, "stress", 1.0),        # cfg 1: 16-wide kernel variant, Gaussian target
    ("mog3", 50, 4, 0.1, 19, "stress", 1.0),       # x_dim 3, three components: run-time-dimension instance, H = 50
    ("mog3", 12, 3, 0.1, 9, "mild", 1.5),          # ... and its 16-wide variant
])
def test_toy_target_loss_gradients_match_autograd(kind, H_nodes, N, eps, B, regime, temp):
    tr, tm, x, z, dx, dz = _small_setup(kind, H_nodes, N, eps, B, regime, temperature=temp)
    loss, x_out, px = tr.calc_loss_and_grads(x, z=z, draws_x=dx, draws_z=dz)
    tt = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))   # noqa: E731
    want, Lx, wpx, Lz, wpz = tm.mog_loss(tt(x), tt(z), tuple(map(tt, dx)), tuple(map(tt, dz)), 0.1)
    want.backward()
    assert H.relerr(tr.last_proposals.cpu().numpy(), np.concatenate([Lx.detach().numpy(), Lz.detach().numpy()])) < 1e-5
    assert np.abs(tr.last_p.cpu().numpy() - np.concatenate([wpx.detach().numpy(), wpz.detach().numpy()])).max() < 2e-5
    want = float(want.detach())
    assert abs(float(loss) - want) <= 2e-4 * max(1., abs(want))
    gv = tr.grad_views()
    worst = {}
    for name, net in (("xnet", tm.xnet), ("vnet", tm.vnet)):
        for k, w in _mlp_packed_ref(net).items():
            got = gv[name][k].cpu().numpy().astype(np.float64).reshape(w.shape)
            worst[f"{name}.{k}"] = float(np.abs(got - w).max() / np.abs(w).max())
    worst["alpha"] = abs(float(gv["alpha"][0]) - float(tm.alpha.grad)) / abs(float(tm.alpha.grad))
    bad = {k: v for k, v in worst.items() if not v <= TOL_G}
    assert not bad, f"gradient mismatch: {bad}\nall: {worst}"
    # accept/reject of the x chains: sampler.py:57-59 (>=)
    acc = (tr.last_p[:B].cpu().numpy() - dx[3]) >= 0
    np.testing.assert_allclose(x_out.cpu().numpy()[acc], tr.last_proposals[:B].cpu().numpy()[acc])
    np.testing.assert_allclose(x_out.cpu().numpy()[~acc], x[~acc].astype(np.float32))


def test_toy_target_training_reduces_loss():
    tr, tm, x, z, dx, dz = _small_setup("mog", 50, 10, 0.1, 256, "init")
    tr.lr_init = 1e-3
    losses = [float(tr.train_step(x, z=z, draws_x=dx, draws_z=dz)[0]) for _ in range(10)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    assert float(tr.dynamics.eps) != pytest.approx(0.1, abs=1e-7)       # alpha is trained
    # the sampling kernel sees the updated weights
    X, V, p = tr.dynamics.forward(x)
    assert torch.isfinite(X).all() and tr.global_step == 10


# ------------------------------------------------------------------ committed fixtures (tests/golden/train_*.npz)
def test_lattice_gradients_match_committed_fixture():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_L8_generic.npz"))
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    T, X, N = int(g["T"]), int(g["X"]), int(g["num_steps"])
    xp, vp = H.gauge_weights(T, X, seed=106, regime=str(g["regime"]))
    dyn = H.gauge_hip(T, X, N, float(g["eps"]), xp, vp, g["masks"], g["x"].shape[0])
    tr = GaugeTrainer(dyn)
    dx = tuple(g[f"draws_x/{i}"] for i in range(4))
    dz = tuple(g[f"draws_z/{i}"] for i in range(4))
    loss, *_ = tr.calc_loss_and_grads(g["x"], float(g["beta"]), z=g["z"], draws_x=dx, draws_z=dz)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=2e-4)
    np.testing.assert_allclose(tr.last_loss_terms.cpu().numpy(), g["terms"], rtol=2e-4,
                               atol=1e-5 * np.abs(g["terms"]).max())
    gv = tr.grad_views()
    assert float(gv["eps"][0]) == pytest.approx(float(g["grad_eps"]), rel=TOL_G)
    D, Hh = 2 * T * X, 4 * 2 * T * X
    for name in ("xnet", "vnet"):
        v = {k: t.cpu().numpy().astype(np.float64) for k, t in gv[name].items()}
        # small tensors are stored in full
        np.testing.assert_allclose(v["b1"], g[f"{name}/v_layer/b"], rtol=0, atol=TOL_G * np.abs(g[f"{name}/v_layer/b"]).max())
        np.testing.assert_allclose(v["bh"], g[f"{name}/h_layer/b"], rtol=0, atol=TOL_G * np.abs(g[f"{name}/h_layer/b"]).max())
        np.testing.assert_allclose(v["wt"], g[f"{name}/t_layer/W"], rtol=0, atol=TOL_G * np.abs(g[f"{name}/t_layer/W"]).max())
        np.testing.assert_allclose(v["coeff_s"], g[f"{name}/coeff_scale"].reshape(-1), rtol=0,
                                   atol=TOL_G * np.abs(g[f"{name}/coeff_scale"]).max())
        np.testing.assert_allclose(v["coeff_q"], g[f"{name}/coeff_transformation"].reshape(-1), rtol=0,
                                   atol=TOL_G * np.abs(g[f"{name}/coeff_transformation"]).max())
        # big matrices: sum |.| and sum of squares of each reference-layout block, and its first row
        blocks = {"v_layer/W": v["w1_t"][:, :D].T, "x_layer/W": v["w1_t"][:, D:].T, "h_layer/W": v["wh_t"].T,
                  "scale_layer/W": v["whd_t"][0].T, "translation_layer/W": v["whd_t"][1].T,
                  "transformation_layer/W": v["whd_t"][2].T}
        for k, blk in blocks.items():
            st = g[f"{name}/{k}/stats"]
            assert np.abs(blk).sum() == pytest.approx(st[1], rel=TOL_G), (name, k)
            assert (blk * blk).sum() == pytest.approx(st[2], rel=2 * TOL_G), (name, k)
            r0 = g[f"{name}/{k}/row0"]
            np.testing.assert_allclose(blk[0], r0, rtol=0, atol=TOL_G * max(np.abs(r0).max(), np.sqrt(st[2] / blk.size)))


def test_toy_target_gradients_match_committed_fixture():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_mog.npz"))
    import l2hmc_amd as la
    from l2hmc_amd.dynamics_trainer import DynamicsTrainer
    N, nh = int(g["trajectory_length"]), int(g["num_nodes"])
    xp, vp = H.mlp_weights(2, nh, seed=106, regime=str(g["regime"]))
    tgt = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    dyn = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=N, eps=float(g["eps"]),
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=nh))
    dyn.set_masks(g["masks"])
    dyn.XNet.load_state(xp)
    dyn.VNet.load_state(vp)
    tr = DynamicsTrainer(dyn, scale=float(g["scale"]))
    dx = tuple(g[f"draws_x/{i}"] for i in range(4))
    dz = tuple(g[f"draws_z/{i}"] for i in range(4))
    loss, x_out, px = tr.calc_loss_and_grads(g["x"], z=g["z"], draws_x=dx, draws_z=dz)
    assert float(loss) == pytest.approx(float(g["loss"]), rel=2e-4)
    assert H.relerr(tr.last_proposals.cpu().numpy(), np.concatenate([g["Lx"], g["Lz"]])) < 1e-5
    gv = tr.grad_views()
    assert float(gv["alpha"][0]) == pytest.approx(float(g["grad_alpha"]), rel=TOL_G)
    for name in ("xnet", "vnet"):
        v = {k: t.cpu().numpy().astype(np.float64) for k, t in gv[name].items()}
        ref = {"w1_t": np.concatenate([g[f"{name}/embed_1/W"], g[f"{name}/embed_2/W"]], axis=0).T,
               "wt": g[f"{name}/embed_3/W"], "b1": g[f"{name}/embed_1/b"], "wh_t": g[f"{name}/linear_1/W"].T,
               "bh": g[f"{name}/linear_1/b"],
               "whd_t": np.stack([g[f"{name}/linear_s/W"].T, g[f"{name}/linear_t/W"].T, g[f"{name}/linear_f/W"].T]),
               "coeff_s": g[f"{name}/scale_s"].reshape(-1), "coeff_q": g[f"{name}/scale_f"].reshape(-1)}
        for k, want in ref.items():
            np.testing.assert_allclose(v[k].reshape(want.shape), want, rtol=0, atol=TOL_G * np.abs(want).max(), err_msg=f"{name}.{k}")


def test_conv3d_gradients_generic_filter_count():
    """num_filters != lattice extent runs the run-time-shaped instantiation of the conv kernels (forward and
    backward); the specialised ones (F = L = 8 / 16) are covered by the tests above."""
    import l2hmc_amd as la
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    from oracle import nets as onets
    T = X = 8
    D, F, N, eps, B = 128, 4, 2, 0.1, 7
    rng = np.random.default_rng(106)
    kw = H.REGIMES["mild"]
    xp = onets.init_conv3d_net(rng, T, D, 2 * D, F, 2., **kw)
    vp = onets.init_conv3d_net(rng, T, D, 2 * D, F, 1., **kw)
    orc = H.gauge_oracle(T, X, N, eps, xp, vp, arch='conv3D')
    dyn = H.gauge_hip(T, X, N, eps, H.conv_weights(T, X, regime="mild")[0], H.conv_weights(T, X, regime="mild")[1],
                      orc.mask, B, arch='conv3D')
    common = dict(_input_shape=(B, T, X, 2), links_shape=(T, X, 2), x_dim=D, spatial_size=X, num_hidden=2 * D,
                  num_filters=F, filter_sizes=[(3, 3, 2), (2, 2, 2)], data_format='channels_last')
    dyn.position_fn = la.ConvNet3D('XNet', factor=2., name_scope='position', **common)
    dyn.momentum_fn = la.ConvNet3D('VNet', factor=1., name_scope='momentum', **common)
    dyn.position_fn.load_state(xp)
    dyn.momentum_fn.load_state(vp)
    tr = GaugeTrainer(dyn)
    tm = TorchGaugeModel(T, X, N, eps, orc.mask, xp, vp, arch='conv3D')
    r2 = np.random.default_rng(7)
    x, z = r2.uniform(0, 2 * np.pi, (B, D)), r2.standard_normal((B, D))
    mk = lambda: (r2.standard_normal((B, D)), r2.standard_normal((B, D)), r2.uniform(size=B), r2.uniform(size=B))  # noqa: E731
    dx, dz = mk(), mk()
    loss, *_ = tr.calc_loss_and_grads(x, 2.5, z=z, draws_x=dx, draws_z=dz)
    want, _ = _ref_grads(tm, x, z, dx, dz, 2.5, 'cos_diff')
    assert abs(float(loss) - want) <= 2e-4 * max(1., abs(want))
    _compare(tr, tm)


def test_toy_target_data_parallel_gradients_equal_full_batch(tmp_path):
    """DynamicsTrainer over two ranks (one GPU, gloo): sharded chains give the full-batch loss and gradient."""
    import socket
    import subprocess
    import sys
    B = 12
    tr, tm, x, z, dx, dz = _small_setup("mog", 50, 5, 0.1, B, "stress")
    loss, *_ = tr.calc_loss_and_grads(x, z=z, draws_x=dx, draws_z=dz)
    full = tr.grads.cpu().numpy().copy()
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    out = str(tmp_path / "dp_toy.npz")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(os.path.dirname(__file__), "dp_train_worker.py"),
                                       out, str(B), "toy"], env=env))
    for p_ in procs:
        assert p_.wait(timeout=300) == 0
    with np.load(out) as f:
        got, got_loss = f["grads"], float(f["loss"])
    assert abs(got_loss - float(loss)) <= 1e-5 * max(1., abs(float(loss)))
    assert np.abs(got - full).max() <= 2e-5 * np.abs(full).max()


def test_large_batch_layered_gradients_equal_fused_gradients():
    """16400 stacked rows: the layered reverse pass now runs its 128-row tiles (gemm_relu_kernel<128, 3|4, .>), the
    fused path its whole-trajectory kernels -- two independent implementations of the same data path must agree
    (the small-batch cases above pin each of them to float64 autograd)."""
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    T = X = 8
    B, N, eps = 8200, 1, 0.1
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    tr = GaugeTrainer(dyn)
    rng = np.random.default_rng(7)
    x, z = rng.uniform(0, 2 * np.pi, (B, 128)), rng.standard_normal((B, 128))
    mk = lambda: (rng.standard_normal((B, 128)), rng.standard_normal((B, 128)), rng.uniform(size=B), rng.uniform(size=B))  # noqa: E731
    dx, dz = mk(), mk()
    res = {}
    for fused in (True, False):
        dyn.fused = fused
        loss, *_ = tr.calc_loss_and_grads(x, 2.0, z=z, draws_x=dx, draws_z=dz)
        res[fused] = (float(loss), {n: {k: v.clone() for k, v in g.items()} if isinstance(g, dict) else g.clone()
                                    for n, g in tr.grad_views().items()})
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * max(1., abs(res[True][0]))
    # Two fp32 implementations with different summation orders: with 16400 x 512 relu gates per layer and call a
    # handful of pre-activations within rounding of 0 fall on different sides, and each such flip moves one row of
    # the hidden-layer gradients by a full term (measured: 3.6e-4 of the largest entry, the same distance either
    # path has from float64 autograd at this size).  So: tight in the Frobenius norm, loose in the max norm.
    for net in ("xnet", "vnet"):
        for k, a in res[True][1][net].items():
            b = res[False][1][net][k]
            assert float((a - b).norm()) <= 1e-4 * float(b.norm()), (net, k)
            assert float((a - b).abs().max()) <= 2e-3 * float(b.abs().max()), (net, k)
    assert abs(float(res[True][1]["eps"][0]) - float(res[False][1]["eps"][0])) <= 1e-4 * abs(float(res[False][1]["eps"][0]))


def test_loss_gradients_on_a_non_square_lattice():
    """4 x 16 (D = 128): the fused taped forward / reverse kernels with T != X, against float64 autograd."""
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    T, X, N, eps, B = 4, 16, 2, 0.1, 11
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    tr = GaugeTrainer(dyn)
    tm = TorchGaugeModel(T, X, N, eps, orc.mask, xp, vp)
    rng = np.random.default_rng(7)
    x, z = rng.uniform(0, 2 * np.pi, (B, 128)), rng.standard_normal((B, 128))
    mk = lambda: (rng.standard_normal((B, 128)), rng.standard_normal((B, 128)), rng.uniform(size=B), rng.uniform(size=B))  # noqa: E731
    dx, dz = mk(), mk()
    loss, *_ = tr.calc_loss_and_grads(x, 2.5, z=z, draws_x=dx, draws_z=dz)
    want, _ = _ref_grads(tm, x, z, dx, dz, 2.5, 'cos_diff')
    assert abs(float(loss) - want) <= 2e-4 * max(1., abs(want))
    _compare(tr, tm)


def test_save_weights_after_training_without_explicit_sync(tmp_path):
    """gauge_model.py:549-554 calls position_fn.save_weights right after training.  The optimiser
    moves only the flat master copy, so every reader of the reference layout must refresh it by itself: weights
    saved WITHOUT trainer.sync_weights() and loaded into a fresh network give the trained S/T/Q."""
    import l2hmc_amd as la
    tr, tm, x, z, dx, dz = _setup(8, 2, 0.15, 16, "init")
    dyn = tr.dynamics
    w_before = dyn.position_fn.state_dict()["h_layer/W"].clone()
    for _ in range(3):
        tr.train_step(x, 2.0, z=z, draws_x=dx, draws_z=dz)
    path = os.path.join(str(tmp_path), "xnet_weights")
    dyn.position_fn.save_weights(path)                   # no sync_weights() call
    fresh = la.GenericNet(model_name='XNet', x_dim=128, num_hidden=512, factor=2., name_scope='position',
                          links_shape=(8, 8, 2))
    fresh.load_weights(path)
    rng = np.random.default_rng(3)
    a, b = rng.standard_normal((9, 128)), rng.uniform(0, 6.3, (9, 128))
    t = np.array([[np.cos(0.3), np.sin(0.3)]])
    for g, w in zip(fresh([a, b, t]), dyn.position_fn([a, b, t])):
        assert H.relerr(g.cpu().numpy(), w.cpu().numpy()) < 1e-6
    assert not torch.equal(w_before, fresh.h_layer.kernel)              # the saved weights are the trained ones
    # the three first-layer biases of the reference layout add up to the packed bias the kernels use
    v = dyn.position_fn.flat_params()[1]
    s = dyn.position_fn.state_dict()
    np.testing.assert_allclose((s["v_layer/b"] + s["x_layer/b"] + s["t_layer/b"]).cpu().numpy(), v["b1"].cpu().numpy(),
                               rtol=0, atol=1e-7)


def test_eps_moves_in_graph_mode_even_if_not_trainable():
    """gauge_model.py:825,965-968 (graph mode) differentiate and apply over dynamics.variables, which holds eps
    also when eps_trainable=False; the eager branch (:820) uses trainable_variables.  Both are reproduced."""
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    for eager, moves in ((False, True), (True, False)):
        tr, tm, x, z, dx, dz = _setup(4, 2, 0.2, 8, "mild")
        dyn = tr.dynamics
        dyn.eps_trainable = False
        tr2 = GaugeTrainer(dyn, lr_init=1e-3, eager_variables=eager)
        e0 = float(dyn.eps)
        tr2.train_step(x, 2.0, z=z, draws_x=dx, draws_z=dz)
        assert (float(dyn.eps) != e0) == moves
