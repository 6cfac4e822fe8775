"""Host-side logic that needs no GPU: weight packing layout, mask / time
construction, chain sharding, golden fixtures vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import nets, lattice as olat, dynamics as ogen
from oracle.gauge_dynamics import make_masks
from l2hmc_amd.network import GenericNet, MLPNet
from l2hmc_amd.dist import shard_bounds
from tests import helpers as H

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_generic_net_packing_layout_matches_reference_dense_layout():
    rng = np.random.default_rng(0)
    D, Hd = 32, 128
    p = nets.init_generic_net(rng, D, Hd, 2., bias_std=0.1, coeff_std=0.1)
    net = GenericNet(model_name='XNet', device=torch.device("cpu"), x_dim=D, num_hidden=Hd, factor=2.,
                     name_scope='position', links_shape=(4, 4, 2))
    net.load_state(p)
    b = {k: v.numpy() for k, v in net._pack_tensors().items()}
    f32 = lambda a: np.asarray(a, dtype=np.float32)  # noqa: E731
    # first input -> v_layer, second -> x_layer (generic_net.py:129-135); k-contiguous rows
    np.testing.assert_array_equal(b["w1_t"][:, :D], f32(p['v_layer/W']).T)
    np.testing.assert_array_equal(b["w1_t"][:, D:], f32(p['x_layer/W']).T)
    np.testing.assert_allclose(b["b1"], f32(p['v_layer/b']) + f32(p['x_layer/b']) + f32(p['t_layer/b']), rtol=1e-6)
    np.testing.assert_array_equal(b["wt"], f32(p['t_layer/W']))
    np.testing.assert_array_equal(b["wh_t"], f32(p['h_layer/W']).T)
    for i, name in enumerate(('scale_layer', 'translation_layer', 'transformation_layer')):
        np.testing.assert_array_equal(b["whd_t"][i], f32(p[name + '/W']).T)
        np.testing.assert_array_equal(b["bhd"][i], f32(p[name + '/b']))
    np.testing.assert_array_equal(b["coeff_s"], f32(p['coeff_scale'])[0])
    # a matmul through the packed form equals the oracle's layer-by-layer form
    v, x = rng.standard_normal((5, D)), rng.standard_normal((5, D))
    h_packed = np.concatenate([v, x], 1) @ b["w1_t"].T.astype(np.float64)
    h_ref = v @ p['v_layer/W'] + x @ p['x_layer/W']
    np.testing.assert_allclose(h_packed, h_ref, atol=1e-5)


def test_reference_initialisation_statistics():
    # generic_net.py:149-161: truncated normal, std = sqrt(1.3 * 2 f / fan_in), zero bias, zero coeffs
    net = GenericNet(model_name='VNet', device=torch.device("cpu"), rng=np.random.RandomState(1), x_dim=128,
                     num_hidden=512, factor=1., name_scope='momentum', links_shape=(8, 8, 2))
    W = net.h_layer.kernel.numpy()
    std = np.sqrt(1.3 * 2. / 512)
    assert np.abs(W).max() <= 2 * std + 1e-7
    assert abs(W.std() / (std * 0.8796) - 1) < 0.02          # std of a 2-sigma truncated normal
    assert net.scale_layer.kernel.abs().max() <= 2 * np.sqrt(1.3 * 0.002 / 512) + 1e-9
    assert float(net.h_layer.bias.abs().sum()) == 0 and float(net.coeff_scale.abs().sum()) == 0
    assert len(net.variables) == 16 and net.q_tanh == 0
    m = MLPNet(2, 'XNet', 2.0, 50, device=torch.device("cpu"))
    assert m.q_tanh == 1 and m.embed_2.kernel.shape == (2, 50) and m.linear_f.kernel.shape == (50, 2)


def test_convnet3d_host_class_matches_reference_shapes():
    from l2hmc_amd.network import ConvNet3D
    net = ConvNet3D('XNet', device=torch.device("cpu"), _input_shape=(4, 8, 8, 2), links_shape=(8, 8, 2), x_dim=128,
                    factor=2., spatial_size=8, num_hidden=256, num_filters=8, filter_sizes=[(3, 3, 2), (2, 2, 2)],
                    name_scope='position', data_format='channels_last')
    # conv_net.py:90-207: 4 conv layers + 7 dense layers + 2 coeff vectors
    assert net.nflat == 64 == nets.conv3d_flat_size(8, 8)
    assert net.conv_x1.kernel.shape == (3, 3, 2, 1, 8) and net.conv_v2.kernel.shape == (2, 2, 2, 8, 16)
    assert net.x_layer.kernel.shape == (64, 256) and net.scale_layer.kernel.shape == (256, 128)
    assert len(net.variables) == 2 + 4 * 2 + 7 * 2
    p = nets.init_conv3d_net(np.random.default_rng(0), 8, 128, 256, 8, 2., bias_std=0.1)
    net.load_state(p)
    np.testing.assert_array_equal(net.state_dict()["conv_v2/W"].numpy(), p["conv_v2/W"].astype(np.float32))
    with pytest.raises(NotImplementedError):
        ConvNet3D('XNet', device=torch.device("cpu"), links_shape=(8, 8, 2), x_dim=128, factor=2., num_hidden=256,
                  num_filters=8, name_scope='position', data_format='channels_first')


def test_mask_and_time_tables_follow_reference_streams():
    np.random.seed(42)
    a = make_masks(10, 128)                       # global legacy stream, gauge_dynamics.py:651-661
    b = make_masks(10, 128, np.random.RandomState(42))
    np.testing.assert_array_equal(a, b)
    np.random.seed(42)
    c = ogen.make_masks(10, 128)                  # utils/dynamics.py:85-96 draws the same stream
    np.testing.assert_array_equal(a, c)
    assert a.sum() == 10 * 64 and set(np.unique(a)) == {0.0, 1.0}


def test_product_constructors_draw_the_reference_mask_stream():
    """Row a5 on the PRODUCT side (the parity tests inject the oracle's masks through set_masks): after
    np.random.seed(42) -- gauge_model.py:195 -- `GaugeDynamics.__init__` (-> _construct_masks_while,
    gauge_dynamics.py:651-661) must put the ones of its first mask row at the reference's indices
    [55, 40, 19, 31, 98, 56, 69, 104, ...] (the frozen legacy NumPy stream), every row holding exactly
    D // 2 ones; `Dynamics._init_mask` (utils/dynamics.py:85-96) draws the same stream for its x_dim."""
    import l2hmc_amd as la
    cpu = torch.device("cpu")
    ref = np.random.RandomState(42)
    want = [ref.permutation(128)[:64] for _ in range(10)]
    assert list(want[0][:8]) == [55, 40, 19, 31, 98, 56, 69, 104]
    for arch in ("generic", "conv3D"):
        np.random.seed(42)
        lat = la.GaugeLattice(8, 8, 2, 'U1', num_samples=4, rand=False)
        dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.25, hmc=False, network_arch=arch, num_steps=10,
                               eps_trainable=True, data_format='channels_last', device=cpu)
        m = dyn.mask.numpy()
        assert m.shape == (10, 128) and m.dtype == np.float32 and set(np.unique(m)) == {0.0, 1.0}
        for s in range(10):
            assert sorted(np.flatnonzero(m[s])) == sorted(want[s]), s
        keep, moved = dyn._get_mask_while(3)
        np.testing.assert_array_equal(keep.numpy(), m[3])
        np.testing.assert_array_equal(moved.numpy(), 1.0 - m[3])
    # the generic integrator: x_dim = 2 -> one of the two coordinates per step, same stream
    np.random.seed(42)
    ref = np.random.RandomState(42)
    want2 = [ref.permutation(2)[:1] for _ in range(10)]
    tgt = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    d2 = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=10, eps=0.1, device=cpu,
                     net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=10, device=cpu))
    m2 = d2.mask.numpy()
    assert m2.shape == (10, 2)
    for s in range(10):
        assert list(np.flatnonzero(m2[s])) == list(want2[s])
        a, b = d2._get_mask(s)
        np.testing.assert_array_equal(a.numpy() + b.numpy(), np.ones(2, np.float32))


def test_ess_estimators_follow_reference_definitions():
    """func_utils.py:45-54,114-120 restated as loops here, vectorised in l2hmc_amd/stats.py."""
    from l2hmc_amd import stats
    rng = np.random.default_rng(0)
    X = rng.standard_normal((12, 5, 3))

    def autocov_ref(X, tau):
        dT, dN, dX = X.shape
        s = 0.
        for t in range(dT - tau):
            s += np.sum(X[t] * X[t + tau]) / dN
        return s / (dT - tau)

    for tau in (0, 1, 5, 11):
        assert abs(stats.autocovariance(X, tau) - autocov_ref(X, tau)) < 1e-12
    A = stats.acl_spectrum(X, 2.0)
    assert A.shape == (11,) and abs(A[3] - autocov_ref(X / 2.0, 3)) < 1e-12
    a = np.array([1.0, 0.5, 0.04, 0.2, -0.3])
    assert abs(stats.ESS(a) - 1. / (1. + 2 * (0.5 + 0.2))) < 1e-12       # entries <= 0.05 are dropped
    # an i.i.d. series has ESS ~ 1 per step once normalised by its variance
    Y = rng.standard_normal((400, 64, 1))
    assert 0.5 < stats.ESS(stats.acl_spectrum(Y, np.sqrt(stats.autocovariance(Y, 0)))) <= 1.0


def test_shard_bounds_partition_the_chains():
    for n, w in ((2048, 8), (8192, 8), (10, 3), (5, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 <= a1
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1


# ------------------------------------------------------------- golden fixtures vs the oracle
def _npz(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def test_golden_u1_observables():
    g = _npz("u1_obs")
    for (T, X) in ((8, 8), (4, 6)):
        k = f"{T}x{X}"
        x = g[k + "/x"]
        np.testing.assert_allclose(olat.plaq_sums(x, T, X), g[k + "/plaq_sums"], atol=1e-13)
        np.testing.assert_allclose(olat.total_action(x, T, X), g[k + "/action"], atol=1e-12)
        np.testing.assert_allclose(2.5 * olat.grad_action(x, T, X), g[k + "/force_beta2.5"], atol=1e-13)
        np.testing.assert_allclose(olat.top_charge(x, T, X), g[k + "/top_charge"], atol=1e-12)


def _gauge_oracle_from_fixture(g):
    T, X, N = int(g["T"]), int(g["X"]), int(g["num_steps"])
    if "xnet/h_layer/W" in g.files:
        xp = {k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith("xnet/")}
        vp = {k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith("vnet/")}
    else:
        arch = str(g["arch"]) if "arch" in g.files else "generic"
        mk = H.conv_weights if arch == "conv3D" else H.gauge_weights
        xp, vp = mk(T, X, seed=106, regime=str(g["regime"]))
        wsum = lambda p: np.array([[np.sum(v), np.sum(np.abs(v))] for _, v in sorted(p.items())])  # noqa: E731
        # the seeded initialiser must reproduce the weights the fixture was made with
        np.testing.assert_allclose(wsum(xp), g["xnet_checksum"], rtol=1e-12)
        np.testing.assert_allclose(wsum(vp), g["vnet_checksum"], rtol=1e-12)
    arch = str(g["arch"]) if "arch" in g.files else "generic"
    orc = H.gauge_oracle(T, X, N, float(g["eps"]), xp, vp, arch=arch)
    np.testing.assert_array_equal(orc.mask, g["masks"])
    return orc, xp, vp


@pytest.mark.parametrize("name", ["gauge_L4_stress", "gauge_L8_cfg3_init", "gauge_L8_cfg3_mild",
                                  "gauge_L8_conv3d_mild"])
def test_golden_gauge_trajectories(name):
    g = _npz(name)
    orc, _, _ = _gauge_oracle_from_fixture(g)
    beta = float(g["beta"])
    trace = []
    orc.transition_kernel(g["x"], beta, g["v0f"], forward=True, trace=trace)
    np.testing.assert_allclose(np.stack([t[0] for t in trace]), g["traj_f/x_steps"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.stack([t[2] for t in trace]), g["traj_f/logdet_steps"], rtol=1e-9, atol=1e-9)
    out = orc.apply_transition(g["x"], beta, g["v0f"], g["v0b"], g["coin"], g["u"])
    for got, key in zip(out, ("x_prop", "v_prop", "p_accept", "x_out")):
        np.testing.assert_allclose(got, g[key], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name,target", [("mog_cfg2", "mog"), ("scg_cfg1", "scg")])
def test_golden_small_targets(name, target):
    g = _npz(name)
    tgt = H.mog_target_oracle() if target == "mog" else H.scg_target_oracle()
    xp = {k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith("xnet/")}
    vp = {k[5:]: g[k].astype(np.float64) for k in g.files if k.startswith("vnet/")}
    orc = ogen.DynamicsOracle(2, tgt, int(g["trajectory_length"]), float(g["eps"]), g["masks"], xp, vp)
    np.testing.assert_allclose(tgt.energy(g["x"]), g["energy"], rtol=1e-10, atol=1e-10)
    Xf, Vf, pf = orc.forward(g["x"], g["v0f"])
    np.testing.assert_allclose(Xf, g["Xf"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(pf, g["pf"], rtol=1e-9, atol=1e-12)
    Lx, Lv, px, outs, _ = ogen.propose(g["x"], orc, g["v0f"], g["v0b"], g["dir_bits"], g["u"], do_mh_step=True)
    np.testing.assert_allclose(outs[0], g["x_accept"], rtol=1e-9, atol=1e-9)


# ---- chain statistics: FFT formulations vs direct lag sums, and a known answer -------
def _ar1(n, walkers, rho, seed=5):
    rng = np.random.default_rng(seed)
    x = np.zeros((n, walkers))
    e = rng.standard_normal((n, walkers))
    for t in range(1, n):
        x[t] = rho * x[t - 1] + e[t]
    return x


def test_autocorr_functions_match_direct_lag_sums():
    from l2hmc_amd import stats
    from oracle import stats as ostats
    x = _ar1(257, 3, 0.6)
    np.testing.assert_allclose(stats.autocorr_func_1d(x[:, 0]), ostats.acf_direct(x[:, 0]), atol=1e-12)
    np.testing.assert_allclose(stats.autocorr_func_1d(x)[:, 2], ostats.acf_direct(x[:, 2]), atol=1e-12)
    np.testing.assert_allclose(stats.autocorr_fast(x[:, 1], kappa=100),
                               ostats.acf_direct(x[:, 1], unbiased=True)[:100], atol=1e-12)
    y = x[:, 0] + 2.0                   # autocorr() does not remove the mean
    want = np.correlate(y, y, mode='full')
    want = (want / want[want.argmax()])[want.size // 2:]
    np.testing.assert_allclose(stats.autocorr(y), want, atol=1e-12)
    tau, curve = stats.calc_iat(x[:, 0], kappa=50)
    assert tau == pytest.approx(1 + 2 * curve.sum())
    assert stats.next_pow_two(1) == 1 and stats.next_pow_two(257) == 512 and stats.next_pow_two(512) == 512
    # torch input (device-resident histories) goes through torch.fft with the same result
    import torch
    np.testing.assert_allclose(stats.autocorr_func_1d(torch.as_tensor(x)), stats.autocorr_func_1d(x), atol=1e-10)


def test_integrated_time_known_answer_and_errors():
    from l2hmc_amd import stats
    from oracle import stats as ostats
    rho = 0.5
    x = _ar1(20000, 8, rho)
    tau, flag = stats.integrated_time(x)
    assert flag is None
    assert tau[0] == pytest.approx((1 + rho) / (1 - rho), rel=0.1)          # AR(1): tau_int = (1+rho)/(1-rho)
    assert tau[0] == pytest.approx(ostats.integrated_time_direct(x[:2000], 5), rel=0.25)
    short = _ar1(300, 4, 0.5)
    assert stats.integrated_time(short, quiet=True)[0][0] == pytest.approx(ostats.integrated_time_direct(short), rel=1e-9)
    with pytest.raises(stats.AutocorrError) as e:
        stats.integrated_time(_ar1(60, 2, 0.9))
    assert e.value.tau.shape == (1,)
    assert stats.integrated_time(_ar1(60, 2, 0.9), quiet=True)[1] == 1
    with pytest.raises(ValueError):
        stats.integrated_time(np.zeros((4, 2, 2, 2)))
    w = _ar1(4000, 6, 0.5).T
    assert stats.autocorr_new(w) == pytest.approx(3.0, rel=0.2)
    assert stats.autocorr_gw2010(w) > 0


def test_block_jackknife_and_run_statistics():
    from l2hmc_amd import stats
    rng = np.random.default_rng(2)
    data = rng.standard_normal(1003)
    blocks = stats.block_resampling(data, 100)
    assert len(blocks) == 100 and all(len(b) in (1003 - 11, 1003 - 10) for b in blocks)
    try:                                    # the reference builds the blocks with sklearn's KFold
        from sklearn.model_selection import KFold
        for (tr, _), b in zip(KFold(n_splits=100).split(data), blocks):
            np.testing.assert_array_equal(data[tr], b)
    except ImportError:
        pass
    avg, err = stats.calc_avg_vals_errors(data, 100)
    assert avg == pytest.approx(data.mean())
    # delete-a-block jackknife of the mean ~ standard error (the reference's formula carries an extra factor
    # num_blocks / (num_blocks - 1) ... times num_blocks; restated as written)
    rs = np.array([b.mean() for b in blocks])
    assert err == pytest.approx(np.sqrt(np.sum((rs - avg) ** 2) / 99 * 100))
    assert len(stats.block_resampling(np.arange(5), 100)) == 5       # fewer samples than blocks
    with pytest.raises(ValueError):
        stats.block_resampling(np.array([]), 3)
    # run statistics on [steps, chains] histories
    steps, chains = 50, 4
    actions, plaqs = rng.uniform(10, 20, (steps, chains)), rng.uniform(0, 1, (steps, chains))
    charges = rng.integers(-2, 3, (steps, chains)).astype(np.float64) + 0.4      # truncation to int as np.array(dtype=int)
    (am, ae), (pm, pe), (qm, qe), (sm, se), probs = stats.calc_observables_stats(actions, plaqs, charges, therm_frac=10)
    np.testing.assert_allclose(am, actions[5:].mean(0))
    np.testing.assert_allclose(pe, plaqs[5:].std(0, ddof=1) / np.sqrt(45))
    qi = charges.astype(int)
    np.testing.assert_allclose(qm, qi[5:].mean(0))
    np.testing.assert_allclose(sm, (qi ** 2).mean(0))                  # not trimmed, as the reference
    assert sum(probs.values()) == pytest.approx(1.0) and set(probs) <= {-1, 0, 1, 2}
    try:
        from scipy.stats import sem as scipy_sem
        np.testing.assert_allclose(stats.sem(actions), scipy_sem(actions))
    except ImportError:
        pass


def test_cpu_baseline_uses_a_sane_thread_count():
    """bench.py's cpu_baseline leg: threads are bounded by the CPUs this process may use (cgroup quota / affinity),
    and torch's thread setting is restored afterwards."""
    import torch
    from oracle.cpu_baseline import effective_cpus, time_cpu_baseline
    from tests import helpers as H
    n = effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    xp, vp = H.gauge_weights(4, 4, regime="init")
    masks = H.gauge_oracle(4, 4, 2, 0.2, xp, vp).mask
    before = torch.get_num_threads()
    r = time_cpu_baseline(4, 4, 2, 0.2, 2.0, 16, xp, vp, masks, budget_s=0.3)
    assert torch.get_num_threads() == before
    assert r["value"] > 0 and 1 <= r["cores"] <= r["cpus_available"] == n and r["calls"] >= 2


def test_step_draw_counter_is_shared_and_never_reused():
    """the native MCMC step takes its Philox stream pair from the dynamics' own draw counter, so a
    sampler started after N dynamics draws (training, apply_transition) shares no stream with them, and two
    samplers on one dynamics interleave instead of repeating each other."""
    from l2hmc_amd._lib import step_draw_index
    used = set()
    draws = 0
    for n_single in (0, 1, 9, 2250, 4):            # single-stream draws (_normal / _uniform) between steps
        for _ in range(n_single):
            assert draws not in used
            used.add(draws)
            draws += 1
        for _ in range(3):                          # three native steps
            d, draws = step_draw_index(draws)
            pair = {2 * d, 2 * d + 1}
            assert not (pair & used) and draws == 2 * d + 2
            used |= pair
    assert step_draw_index(0) == (0, 2) and step_draw_index(1) == (1, 4) and step_draw_index(2) == (1, 4)


def test_reference_keyed_views_of_run_and_train_histories():
    """the reference pickles dicts keyed by (step, beta) (gauge_model.py:1196-1205,
    :1409-1413); the helpers re-key this package's array histories the same way."""
    from l2hmc_amd.gauge_trainer import GaugeTrainer
    from l2hmc_amd.gauge_sampler import GaugeSampler
    steps, chains = 3, 4
    out = {k: np.arange(steps * chains, dtype=np.float64).reshape(steps, chains) + i
           for i, k in enumerate(("actions", "plaqs", "charges"))}
    out.update(loss=np.array([1., 2., 3.]), accept_prob=np.array([.1, .2, .3]), charge_diff=np.array([0., 1., 0.]),
               beta=np.array([2.0, 2.1, 2.2]))
    d = GaugeTrainer.train_data_dict(out, initial_step=10)
    assert set(d) == {'loss', 'actions', 'plaqs', 'charges', 'charge_diff', 'accept_prob'}
    assert list(d['loss']) == [(10, 2.0), (11, 2.1), (12, 2.2)] and d['loss'][(11, 2.1)] == 2.0
    np.testing.assert_array_equal(d['plaqs'][(12, 2.2)], out['plaqs'][2])
    run = dict(px=np.zeros((steps, chains)), actions=out["actions"], plaqs=out["plaqs"], charges=out["charges"],
               charge_diff=np.zeros((steps, chains)))
    a, p, q, dq = GaugeSampler.run_dicts(run, 4.0)
    assert list(a) == [(0, 4.0), (1, 4.0), (2, 4.0)]
    np.testing.assert_array_equal(q[(1, 4.0)], out["charges"][1])
