"""GPU parity tests: every call goes through the C ABI of libl2hmc_hip.so and is
compared with the CPU oracle on identical injected inputs.

Tolerances.  north_star: "within 1e-5 relative fp32".  `relerr` is the max
abs deviation relative to the tensor's own scale (>= 1).
  * single ops and single leapfrog steps:           TOL_OP  = 1e-5 (measured ~2e-7..2e-6)
  * accept probabilities (O(1), exp of an O(100) energy difference in fp32):  TOL_P = 2e-5 abs
  * whole trajectories: a hot-start trajectory at eps=0.25 amplifies rounding
    ~100x over 10 steps (single steps agree to ~2e-7, the fp32 floor), so even
    the reference's own fp32 graph is ~1e-5 away from exact arithmetic at the
    end: the fp32 NumPy oracle (same op order as the reference) measures that
    intrinsic deviation on the same inputs.  The allowance over it is MEASURED
    (tools/error_ratio.py -> profiles/r02_error_ratio_*.txt; per leapfrog step,
    x / v / log-det / p, fused and layered, cfg 3 / cfg 3 conv / cfg 4 dynamics):
      - averaged over samples the HIP error is 0.7-1.1 x the fp32 oracle's for
        x and v at every step (it is usually SMALLER: fp64 energy differences,
        fused multiply-adds), <= 1.4 x for the log-det while that is < 1e-6;
      - RMS over a sample's elements: <= 1.0 x where it binds at cfg 3, worst
        single sample 1.53 x (cfg 4, 16 chains)       -> RMS_RATIO = 1.6;
      - the 99.9 % quantile of the element-wise error (a tail statistic that is
        not a single extreme value): worst single sample 2.09 x over the
        trajectories of all four tables                   -> Q999_RATIO = 2.5
        (applied to samples of >= 4000 elements);
      - the MAX over a sample's ~10^4 chaotically amplified elements is an
        extreme value of a heavy-tailed distribution on both sides: the ratio of
        the two maxima scatters (worst single sample 4.61 among 96 32-chain
        trajectories, 3.62 among 32 128-chain ones, typical 0.7-1.3; the
        libm-exp/tanh diagnostic build has the same means) -> MAX_RATIO = 5 is
        an extreme-value allowance, not a precision one;
      - accept probability: a few chains per sample have p != 0, so the max
        ratio scatters most (worst 2.74, typical <= 2.2) -> P_RATIO = 3.5.
    Both apply only above the absolute bars (TOL_OP max, TOL_OP / 3 RMS, TOL_P):
    on benign dynamics (small step, near-cold start) the whole trajectory is
    held to TOL_OP outright (test_trajectory_within_1e5_on_benign_dynamics).
"""
import os

import numpy as np
import pytest
import torch

from oracle import lattice as olat, nets as onets, dynamics as ogen
from tests import helpers as H

pytestmark = pytest.mark.gpu

TOL_OP = 1e-5
TOL_P = 2e-5
MAX_RATIO, Q999_RATIO, RMS_RATIO, P_RATIO = 5.0, 2.5, 1.6, 3.5     # measured allowances over the fp32 oracle's own error
FORMS_TOL = 1e-4                                         # toy kernel's two first-layer forms (see the test)
FUSED_VS_LAYERED_X, FUSED_VS_LAYERED_P = 1.5e-5, 2e-5    # full-size cfg 3, fused against layered: measured 4.4e-6 / 0
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H_REG = H.REGIMES


def np_(t):
    return t.detach().cpu().numpy().astype(np.float64)


def rmserr(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.sqrt(np.mean((got - want) ** 2)) / max(1.0, np.max(np.abs(want))))


def assert_fp32_equivalent(got, want64, want32, what):
    """`got` is as close to the fp64 oracle as an fp32 evaluation in the reference's op order."""
    emax, imax = H.relerr(got, want64), H.relerr(want32, want64)
    erms, irms = rmserr(got, want64), rmserr(want32, want64)
    assert emax < max(TOL_OP, MAX_RATIO * imax), f"{what}: max err {emax:.2e} vs intrinsic fp32 {imax:.2e}"
    assert erms < max(TOL_OP / 3, RMS_RATIO * irms), f"{what}: rms err {erms:.2e} vs intrinsic fp32 {irms:.2e}"
    if np.size(want64) >= 4000:
        scale = max(1.0, np.max(np.abs(want64)))
        eq = np.quantile(np.abs(np.asarray(got, dtype=np.float64) - want64), 0.999) / scale
        iq = np.quantile(np.abs(np.asarray(want32, dtype=np.float64) - want64), 0.999) / scale
        assert eq < max(TOL_OP / 2, Q999_RATIO * iq), f"{what}: 99.9 % quantile {eq:.2e} vs intrinsic fp32 {iq:.2e}"


@pytest.fixture(scope="module")
def la():
    import l2hmc_amd
    from l2hmc_amd import _lib
    _lib.lib()
    return l2hmc_amd


def test_native_library_is_the_one_loaded(la):
    maps = open("/proc/self/maps").read()
    assert "l2hmc_amd/libl2hmc_hip.so" in maps


# ----------------------------------------------------------------- lattice
# every lanes-per-chain class of the two-sites-per-thread kernel (32: 8x8, 4x16, 32x2; 64: 8x16; 128: 16x16; 256: 16x32,
# 32x16; 512: 32x32, 16x64), ragged batches (partial workgroups), and shapes that take the general kernel
@pytest.mark.parametrize("T,X,B", [(8, 8, 7), (8, 8, 1), (16, 16, 5), (32, 32, 3), (4, 6, 9), (17, 17, 2), (4, 4, 300),
                                   (8, 16, 5), (16, 32, 3), (32, 16, 2), (4, 16, 11), (32, 2, 9), (16, 64, 2),
                                   (8, 8, 1027), (64, 1, 3), (5, 8, 4)])
def test_u1_action_force_observables(la, T, X, B):
    x = np.random.default_rng(1).uniform(-7, 7, (B, 2 * T * X)).astype(np.float32)
    o = la.u1_observables(x, T, X, beta=2.5, want_force=True)
    x64 = x.astype(np.float64)
    assert H.relerr(np_(o["action"]), olat.total_action(x64, T, X)) < TOL_OP
    assert H.relerr(np_(o["force"]), 2.5 * olat.grad_action(x64, T, X)) < TOL_OP
    assert H.relerr(np_(o["avg_plaq"]), olat.avg_plaq(x64, T, X)) < TOL_OP
    assert H.relerr(np_(o["top_charge"]), olat.top_charge(x64, T, X)) < TOL_OP
    lat = la.GaugeLattice(T, X, 2, 'U1', num_samples=B, rand=False)
    assert H.relerr(np_(lat.calc_plaq_sums(x)), olat.plaq_sums(x64, T, X)) < TOL_OP


def test_u1_known_answers_and_edge_cases(la):
    lat = la.GaugeLattice(8, 8, 2, 'U1', num_samples=5, rand=False)
    S, plaq, Q = lat.calc_plaq_observables(lat.samples.reshape(5, -1))      # cold start: notebook step-0 row
    assert torch.all(S == 0) and torch.all(plaq == 1) and torch.all(Q == 0)
    assert torch.all(lat.grad_action(lat.samples.reshape(5, -1), 2.0) == 0)
    # pure gauge transformation leaves every plaquette invariant
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 2 * np.pi, (3, 8, 8, 2))
    lam = rng.uniform(0, 2 * np.pi, (3, 8, 8))
    xg = x.copy()
    xg[..., 0] += lam - np.roll(lam, -1, axis=1)      # link 0 runs along i (x1[i+1,j] closes the plaquette)
    xg[..., 1] += lam - np.roll(lam, -1, axis=2)      # link 1 runs along j
    a = la.u1_observables(x.reshape(3, -1), 8, 8)["action"]
    b = la.u1_observables(xg.reshape(3, -1), 8, 8)["action"]
    assert H.relerr(np_(b), np_(a)) < 1e-5
    # empty batch
    e = la.u1_observables(np.zeros((0, 128), np.float32), 8, 8, want_force=True)
    assert e["action"].shape == (0,) and e["force"].shape == (0, 128)
    with pytest.raises(NotImplementedError):
        la.GaugeLattice(8, 8, 2, 'SU2', num_samples=2)


def test_golden_u1(la):
    g = np.load(os.path.join(GOLD, "u1_obs.npz"))
    for (T, X) in ((8, 8), (4, 6)):
        k = f"{T}x{X}"
        o = la.u1_observables(g[k + "/x"], T, X, beta=2.5, want_force=True)
        assert H.relerr(np_(o["action"]), g[k + "/action"]) < TOL_OP
        assert H.relerr(np_(o["force"]), g[k + "/force_beta2.5"]) < TOL_OP
        assert H.relerr(np_(o["top_charge"]), g[k + "/top_charge"]) < TOL_OP


# ----------------------------------------------------------------- dense S/T/Q net
@pytest.mark.parametrize("regime", ["init", "stress"])
@pytest.mark.parametrize("D,rows", [(128, 100), (128, 1), (32, 65), (512, 70),
                                    (128, 5003),        # 64 x 128 tiles with 32-deep k-tiles (grid > 256 workgroups)
                                    (128, 16411),       # >= 512 tiles of 128 x 128: the large-grid instantiation
                                    (128, 32771)])      # >= 512 tiles of 128 x 64 x 3 heads: heads32_kernel (32x32x2 form)
def test_stq_dense_matches_generic_net(la, regime, D, rows):
    rng = np.random.default_rng(5)
    p = onets.init_generic_net(np.random.default_rng(106), D, 4 * D, 2., **H.REGIMES[regime])
    net = la.GenericNet(model_name='XNet', x_dim=D, num_hidden=4 * D, factor=2., name_scope='position',
                        links_shape=(1, D // 2, 2))
    net.load_state(p)
    a, b = rng.standard_normal((rows, D)), rng.uniform(0, 6.3, (rows, D))
    t = np.array([[np.cos(0.7), np.sin(0.7)]])
    S, T, Q = net([a, b, t])
    # the oracle gets the fp32-rounded inputs the kernel sees
    a32, b32 = a.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    p32 = {k: v.astype(np.float32).astype(np.float64) for k, v in p.items()}
    So, To, Qo = onets.generic_net(p32, [a32, b32, np.tile(t.astype(np.float32).astype(np.float64), (rows, 1))])
    assert H.relerr(np_(S), So) < TOL_OP and H.relerr(np_(T), To) < TOL_OP and H.relerr(np_(Q), Qo) < TOL_OP
    assert np.abs(So).max() > (0.1 if regime == "stress" else 1e-3)      # the outputs are not trivially zero


@pytest.mark.parametrize("D,Hn,rows", [(24, 96, 4), (72, 288, 70), (30, 120, 9), (50, 77, 131)])
def test_stq_dense_any_width(la, D, Hn, rows):
    """generic_net.py:20-93 takes any x_dim / num_hidden: widths that are not multiples of 32 (a 6x6 lattice has
    x_dim 72, H 288) and rows that are not 16-byte aligned (x_dim 30, 50) run the RAGGED instantiation of the
    layered kernels -- same arithmetic on zero-padded k-tiles."""
    rng = np.random.default_rng(5)
    p = onets.init_generic_net(np.random.default_rng(106), D, Hn, 2., **H_REG["stress"])
    net = la.GenericNet(model_name='XNet', x_dim=D, num_hidden=Hn, factor=2., name_scope='position',
                        links_shape=(1, D // 2, 2))
    net.load_state(p)
    a, b = rng.standard_normal((rows, D)), rng.uniform(0, 6.3, (rows, D))
    t = np.array([[np.cos(0.7), np.sin(0.7)]])
    S, T, Q = net([a, b, t])
    f32 = lambda z: z.astype(np.float32).astype(np.float64)      # noqa: E731
    p32 = {k: f32(v) for k, v in p.items()}
    So, To, Qo = onets.generic_net(p32, [f32(a), f32(b), np.tile(f32(t), (rows, 1))])
    assert H.relerr(np_(S), So) < TOL_OP and H.relerr(np_(T), To) < TOL_OP and H.relerr(np_(Q), Qo) < TOL_OP
    assert np.abs(So).max() > 0.05


@pytest.mark.parametrize("T,X", [(6, 6), (3, 5), (2, 4)])
def test_dynamics_on_lattices_whose_widths_are_not_multiples_of_32(la, T, X):
    """The whole operator surface on odd shapes (x_dim 72, 30, 16): leapfrog steps, both trajectories and
    apply_transition against the oracle -- gauge_dynamics.py:169-187 builds num_hidden = 4 * x_dim for any lattice."""
    N, eps, beta, B, D = 3, 0.15, 2.5, 11, 2 * T * X
    xp, vp = H.gauge_weights(T, X, regime="stress")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D)
    for step in (0, N - 1):
        for fn, ofn in ((dyn._forward_lf, orc._forward_lf), (dyn._backward_lf, orc._backward_lf)):
            x1, v1, ld = fn(x, v0f, beta, step)
            ox, ov, old = ofn(x, v0f, beta, step)
            assert H.relerr(np_(x1), ox) < TOL_OP and H.relerr(np_(v1), ov) < TOL_OP and H.relerr(np_(ld), old) < TOL_OP
    want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    for both in (True, False):
        dyn.both_directions = both
        got = [np_(g) for g in dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)]
        assert H.relerr(got[0], want[0]) < 2 * TOL_OP and H.relerr(got[1], want[1]) < 2 * TOL_OP
        assert np.abs(got[2] - want[2]).max() < TOL_P
    smp = la.GaugeSampler(dyn)                       # native MCMC step on the same shape (general path)
    xs = torch.as_tensor(x, dtype=torch.float32, device="cuda")
    xn, px, obs, dq = smp.step(xs, beta)
    assert xn.shape == xs.shape and float(xn.min()) >= 0 and torch.all((px >= 0) & (px <= 1))
    assert H.relerr(np_(obs["action"]), olat.total_action(np_(xs), T, X)) < TOL_OP


# ----------------------------------------------------------------- gauge dynamics
def _pair(T, X, N, eps, B, regime, fused=True, hmc=False, both=True):
    xp, vp = H.gauge_weights(T, X, regime=regime)
    orc = H.gauge_oracle(T, X, N, eps, xp, vp, hmc=hmc)
    orc32 = H.gauge_oracle(T, X, N, eps, xp, vp, hmc=hmc, dtype=np.float32)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B, hmc=hmc, both_directions=both)
    dyn.fused = fused
    return orc, orc32, dyn


CASES = [  # T, X, N, eps, beta, B, regime, fused
    (8, 8, 2, 0.25, 2.0, 1, "stress", True),      # a single chain (15 of the workgroup's 16 rows are padding)
    (8, 8, 10, 0.25, 2.0, 71, "init", True),      # cfg-3 shape, whole-trajectory kernel, ragged batch
    (8, 8, 10, 0.25, 2.0, 71, "init", False),     # same through the layer-by-layer kernels
    (8, 8, 10, 0.25, 2.0, 33, "mild", True),
    (4, 16, 4, 0.2, 2.0, 19, "mild", True),       # non-square lattice with x_dim 128: still the fused kernel
    (8, 8, 3, 0.2, 2.5, 16, "stress", True),
    (8, 8, 3, 0.2, 2.5, 17, "stress", False),
    (4, 4, 3, 0.2, 2.5, 10, "stress", True),      # D=32: no fused kernel for this shape -> layered path
    (16, 16, 2, 0.1, 3.0, 5, "init", True),       # D=512, H=2048
]


def test_leapfrog_step_on_a_chip_filling_grid(la):
    """32771 rows of the 8x8 lattice through the layer-by-layer kernels: the grid sizes at which the large-tile
    instantiations run (gemm_relu_kernel<128, .>, heads32_kernel with its fused sub-updates and log-det partial sums),
    ragged last tile included; every row against the oracle."""
    T = X = 8
    N, eps, beta, B = 4, 0.2, 2.5, 32771
    orc, _, dyn = _pair(T, X, N, eps, B, "stress", False)
    rng = np.random.default_rng(77)
    x, v = rng.uniform(0, 2 * np.pi, (B, 128)), rng.standard_normal((B, 128))
    for step, lf, olf in ((1, dyn._forward_lf, orc._forward_lf), (2, dyn._backward_lf, orc._backward_lf)):
        x1, v1, ld = lf(x, v, beta, step)
        ox, ov, old = olf(x.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64), beta, step)
        assert H.relerr(np_(x1), ox) < TOL_OP and H.relerr(np_(v1), ov) < TOL_OP and H.relerr(np_(ld), old) < TOL_OP


@pytest.mark.parametrize("T,X,N,eps,beta,B,regime,fused", CASES)
def test_leapfrog_step_and_its_inverse(la, T, X, N, eps, beta, B, regime, fused):
    orc, _, dyn = _pair(T, X, N, eps, B, regime, fused)
    x, v0f, _, _, _ = H.gauge_inputs(B, 2 * T * X)
    for step in (0, N - 1):
        x1, v1, ld = dyn._forward_lf(x, v0f, beta, step)
        ox, ov, old = orc._forward_lf(x, v0f, beta, step)
        assert H.relerr(np_(x1), ox) < TOL_OP and H.relerr(np_(v1), ov) < TOL_OP and H.relerr(np_(ld), old) < TOL_OP
        xb, vb, ldb = dyn._backward_lf(x, v0f, beta, step)
        ox, ov, old = orc._backward_lf(x, v0f, beta, step)
        assert H.relerr(np_(xb), ox) < TOL_OP and H.relerr(np_(vb), ov) < TOL_OP and H.relerr(np_(ldb), old) < TOL_OP
        # gauge_dynamics.py:537-590: backward inverts forward (index reversed inside _backward_lf)
        x2, v2, ld2 = dyn._backward_lf(x1, v1, beta, N - 1 - step)
        assert H.relerr(np_(x2), x) < 2e-5 and H.relerr(np_(v2), v0f) < 2e-5
        assert H.relerr(np_(ld2), -np_(ld)) < 2e-5


@pytest.mark.parametrize("T,X,N,eps,beta,B,regime,fused", CASES)
def test_transition_kernel_both_directions(la, T, X, N, eps, beta, B, regime, fused):
    orc, orc32, dyn = _pair(T, X, N, eps, B, regime, fused)
    x, v0f, v0b, _, _ = H.gauge_inputs(B, 2 * T * X)
    for fwd, v0 in ((True, v0f), (False, v0b)):
        xo, vo, p, sld = dyn.transition_kernel(x, beta, forward=fwd, momentum=v0, return_logdet=True)
        want = orc.transition_kernel(x, beta, v0, forward=fwd)
        f32 = orc32.transition_kernel(x.astype(np.float32), beta, v0.astype(np.float32), forward=fwd)
        assert_fp32_equivalent(np_(xo), want[0], f32[0], "x")
        assert_fp32_equivalent(np_(vo), want[1], f32[1], "v")
        assert_fp32_equivalent(np_(sld), want[3], f32[3], "sumlogdet")
        assert np.abs(np_(p) - want[2]).max() < max(TOL_P, P_RATIO * np.abs(f32[2] - want[2]).max())


def test_trajectory_within_1e5_on_benign_dynamics(la):
    """Small step, near-cold start: the 1e-5 bar holds for the whole 10-step trajectory outright."""
    T = X = 8
    N, eps, beta, B = 10, 0.1, 2.0, 40
    orc, _, dyn = _pair(T, X, N, eps, B, "mild")
    rng = np.random.default_rng(7)
    x = rng.normal(0, 0.3, (B, 128))
    v = rng.standard_normal((B, 128))
    for fwd in (True, False):
        xo, vo, p, sld = dyn.transition_kernel(x, beta, forward=fwd, momentum=v, return_logdet=True)
        want = orc.transition_kernel(x, beta, v, forward=fwd)
        assert H.relerr(np_(xo), want[0]) < TOL_OP and H.relerr(np_(vo), want[1]) < TOL_OP
        assert np.abs(np_(p) - want[2]).max() < TOL_P and H.relerr(np_(sld), want[3]) < TOL_OP
        assert want[2].mean() > 0.05       # a regime where proposals actually get accepted


@pytest.mark.parametrize("T,X,N,eps,beta,B,regime,fused", CASES[:8])
def test_apply_transition_matches_oracle_in_both_modes(la, T, X, N, eps, beta, B, regime, fused):
    orc, orc32, dyn = _pair(T, X, N, eps, B, regime, fused)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, 2 * T * X)
    want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    f32 = orc32.apply_transition(x.astype(np.float32), beta, v0f.astype(np.float32), v0b.astype(np.float32),
                                 coin, u.astype(np.float32))
    outs = {}
    for both in (True, False):
        dyn.both_directions = both
        got = dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
        outs[both] = [np_(g) for g in got]
        assert_fp32_equivalent(outs[both][0], want[0], f32[0], "x_prop")
        assert_fp32_equivalent(outs[both][1], want[1], f32[1], "v_prop")
        assert np.abs(outs[both][2] - want[2]).max() < max(TOL_P, P_RATIO * np.abs(f32[2] - want[2]).max())
        safe = np.abs(want[2] - u) > 1e-4        # accept decisions can only flip where p - u is within rounding
        assert_fp32_equivalent(outs[both][3][safe], want[3][safe], f32[3][safe], "x_out")
        acc = outs[both][2] > u
        np.testing.assert_array_equal(outs[both][3][acc & safe], outs[both][0][acc & safe])
        np.testing.assert_array_equal(outs[both][3][~acc & safe], x.astype(np.float32)[~acc & safe])
    # selected-direction mode returns exactly what both-directions mode returns
    for a, b in zip(outs[True], outs[False]):
        np.testing.assert_array_equal(a, b)


def test_hmc_mode_is_plain_leapfrog(la):
    T = X = 8
    N, eps, beta, B = 5, 0.05, 2.0, 20
    orc, _, dyn = _pair(T, X, N, eps, B, "init", hmc=True)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, 128)
    xo, vo, p, sld = dyn.transition_kernel(x, beta, forward=True, momentum=v0f, return_logdet=True)
    want = orc.transition_kernel(x, beta, v0f, forward=True)
    assert H.relerr(np_(xo), want[0]) < TOL_OP and H.relerr(np_(vo), want[1]) < TOL_OP
    assert torch.all(sld == 0) and np.abs(np_(p) - want[2]).max() < TOL_P, np.abs(np_(p) - want[2]).max()
    got = dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
    w = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    assert H.relerr(np_(got[0]), w[0]) < TOL_OP and np.abs(np_(got[2]) - w[2]).max() < TOL_P, \
        np.abs(np_(got[2]) - w[2]).max()


def test_sub_update_methods_on_materialised_stq(la):
    """The reference's public sub-update methods (gauge_dynamics.py:486-590) through the standalone ops."""
    T = X = 8
    orc, _, dyn = _pair(T, X, 4, 0.2, 12, "stress")
    x, v, _, _, _ = H.gauge_inputs(12, 128)
    t = dyn._format_time(1, tile=12)
    to = orc._format_time(1, tile=12)
    assert H.relerr(np_(t), to) < 1e-6
    m, mi = dyn._get_mask_while(1)
    mo, mio = orc._get_mask_while(1)
    for fn, ofn, args, oargs in (
            (dyn._update_momentum_forward, orc._update_momentum_forward, (x, v, 2.0, t), (x, v, 2.0, to)),
            (dyn._update_momentum_backward, orc._update_momentum_backward, (x, v, 2.0, t), (x, v, 2.0, to)),
            (dyn._update_position_forward, orc._update_position_forward, (x, v, t, m, mi), (x, v, to, mo, mio)),
            (dyn._update_position_backward, orc._update_position_backward, (x, v, t, mi, m), (x, v, to, mio, mo))):
        got, ld = fn(*args)
        want, old = ofn(*oargs)
        assert H.relerr(np_(got), want) < TOL_OP and H.relerr(np_(ld), old) < TOL_OP
    h = dyn.hamiltonian(x, v, 2.0)
    assert H.relerr(np_(h), orc.hamiltonian(x, v, 2.0)) < TOL_OP
    p = dyn._compute_accept_prob(x, v, x, v, np.zeros(12), 2.0)
    assert torch.all(p == 1)
    # non-finite -> 0 (gauge_dynamics.py:609)
    bad = dyn._compute_accept_prob(x, v, x, v, np.full(12, np.nan), 2.0)
    assert torch.all(bad == 0)


def test_constructor_surface_and_errors(la):
    lat = la.GaugeLattice(8, 8, 2, 'U1', num_samples=4, rand=True)
    fn = lat.get_energy_function()
    d = la.GaugeDynamics(lat, fn, eps=0.3, hmc=False, network_arch='generic', num_steps=3, eps_trainable=True,
                         data_format='channels_last')
    assert d.batch_size == 4 and d.x_dim == 128 and abs(float(d.eps) - 0.3) < 1e-7
    assert d.mask.shape == (3, 128) and torch.all(d.mask.sum(1) == 64)
    assert len(d.trainable_variables) == 1 + 2 * 16 and d.position_fn.num_hidden == 512
    out = d(torch.as_tensor(lat.samples.reshape(4, -1)), 2.0)
    assert [tuple(o.shape) for o in out] == [(4, 128), (4, 128), (4,), (4, 128)]
    assert torch.all((out[2] >= 0) & (out[2] <= 1))
    with pytest.raises(NotImplementedError):  # conv2D is out of scope
        la.GaugeDynamics(lat, fn, eps=0.3, hmc=False, network_arch='conv2D', num_steps=3, eps_trainable=True)
    with pytest.raises(AttributeError):      # gauge_dynamics.py:117-119
        la.GaugeDynamics(lat, fn, eps=0.3, hmc=False, network_arch='bogus', num_steps=3, eps_trainable=True)
    with pytest.raises(NotImplementedError):
        la.GaugeDynamics(lat, lambda x: x.sum(1), eps=0.3, hmc=True, num_steps=3, eps_trainable=True)
    d.position_fn.save_weights("/tmp/_xnet_weights")
    d.position_fn.load_weights("/tmp/_xnet_weights")


# ----------------------------------------------------------------- ConvNet3D (network/conv_net.py)
@pytest.mark.parametrize("L,rows,regime", [(8, 37, "stress"), (8, 3, "init"), (16, 9, "stress")])
def test_stq_conv3d_matches_oracle(la, L, rows, regime):
    D = 2 * L * L
    xp, _ = H.conv_weights(L, L, regime=regime)
    net = la.ConvNet3D('XNet', _input_shape=(rows, L, L, 2), links_shape=(L, L, 2), x_dim=D, factor=2.,
                       spatial_size=L, num_hidden=2 * D, num_filters=L, filter_sizes=[(3, 3, 2), (2, 2, 2)],
                       name_scope='position', data_format='channels_last')
    net.load_state(xp)
    rng = np.random.default_rng(5)
    a, b = rng.standard_normal((rows, D)), rng.uniform(0, 6.3, (rows, D))
    t = np.array([[np.cos(0.7), np.sin(0.7)]])
    S, Tr, Q = net([a, b, t])
    p32 = {k: v.astype(np.float32).astype(np.float64) for k, v in xp.items()}
    f32 = lambda z: z.astype(np.float32).astype(np.float64)   # noqa: E731
    So, To, Qo = onets.conv3d_net(p32, [f32(a), f32(b), np.tile(f32(t), (rows, 1))], (L, L, 2))
    assert H.relerr(np_(S), So) < TOL_OP and H.relerr(np_(Tr), To) < TOL_OP and H.relerr(np_(Q), Qo) < TOL_OP
    assert np.abs(To).max() > 1e-4


@pytest.mark.parametrize("L", [8, 16])
def test_stq_conv3d_with_inputs_that_are_not_16_byte_aligned(la, L):
    """The compile-time front-end instances stage the chains with 16-byte loads; inputs carved out of a buffer at an odd
    float offset must take the run-time form (launch_conv3d_front) and still match the oracle."""
    rows, D = 5, 2 * L * L
    xp, _ = H.conv_weights(L, L, regime="stress")
    net = la.ConvNet3D('XNet', _input_shape=(rows, L, L, 2), links_shape=(L, L, 2), x_dim=D, factor=2.,
                       spatial_size=L, num_hidden=2 * D, num_filters=L, filter_sizes=[(3, 3, 2), (2, 2, 2)],
                       name_scope='position', data_format='channels_last')
    net.load_state(xp)
    rng = np.random.default_rng(11)
    a, b = rng.standard_normal((rows, D)), rng.uniform(0, 6.3, (rows, D))
    t = np.array([[np.cos(0.4), np.sin(0.4)]])

    def odd(z):          # the same values at data_ptr % 16 == 4
        buf = torch.empty(rows * D + 1, dtype=torch.float32, device="cuda")
        v = buf[1:].view(rows, D)
        v.copy_(torch.as_tensor(z, dtype=torch.float32))
        assert v.data_ptr() % 16 == 4 and v.is_contiguous()
        return v

    S, Tr, Q = net([odd(a), odd(b), t])
    S2, T2, Q2 = net([a, b, t])                                # aligned: the compile-time instance
    p32 = {k: v.astype(np.float32).astype(np.float64) for k, v in xp.items()}
    f32 = lambda z: z.astype(np.float32).astype(np.float64)   # noqa: E731
    So, To, Qo = onets.conv3d_net(p32, [f32(a), f32(b), np.tile(f32(t), (rows, 1))], (L, L, 2))
    for got, ref in ((S, So), (Tr, To), (Q, Qo), (S2, So), (T2, To), (Q2, Qo)):
        assert H.relerr(np_(got), ref) < TOL_OP


@pytest.mark.parametrize("L,N,B,regime", [(8, 5, 37, "mild"), (8, 3, 16, "stress"), (16, 2, 6, "init")])
def test_conv3d_dynamics_matches_oracle(la, L, N, B, regime):
    eps, beta, D = 0.2, 2.0, 2 * L * L
    xp, vp = H.conv_weights(L, L, regime=regime)
    orc = H.gauge_oracle(L, L, N, eps, xp, vp, arch='conv3D')
    orc32 = H.gauge_oracle(L, L, N, eps, xp, vp, arch='conv3D', dtype=np.float32)
    dyn = H.gauge_hip(L, L, N, eps, xp, vp, orc.mask, B, arch='conv3D')
    assert dyn.position_fn.num_hidden == 2 * D and dyn.position_fn.num_filters == L     # gauge_dynamics.py:129-130
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D)
    for step in (0, N - 1):
        for fn, ofn in ((dyn._forward_lf, orc._forward_lf), (dyn._backward_lf, orc._backward_lf)):
            x1, v1, ld = fn(x, v0f, beta, step)
            ox, ov, old = ofn(x, v0f, beta, step)
            assert H.relerr(np_(x1), ox) < TOL_OP and H.relerr(np_(v1), ov) < TOL_OP
            assert H.relerr(np_(ld), old) < TOL_OP
    want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    f32 = orc32.apply_transition(x.astype(np.float32), beta, v0f.astype(np.float32), v0b.astype(np.float32), coin,
                                 u.astype(np.float32))
    for both, fused in ((True, True), (False, True), (True, False)):      # fused whole-trajectory kernel at L=8
        dyn.both_directions, dyn.fused = both, fused
        got = [np_(g) for g in dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)]
        assert_fp32_equivalent(got[0], want[0], f32[0], "x_prop")
        assert_fp32_equivalent(got[1], want[1], f32[1], "v_prop")
        assert np.abs(got[2] - want[2]).max() < max(TOL_P, P_RATIO * np.abs(f32[2] - want[2]).max())
    dyn.fused = True


# ----------------------------------------------------------------- golden fixtures
@pytest.mark.parametrize("name,fused", [("gauge_L4_stress", True), ("gauge_L8_cfg3_init", True),
                                        ("gauge_L8_cfg3_init", False), ("gauge_L8_cfg3_mild", True),
                                        ("gauge_L8_conv3d_mild", True)])
def test_golden_gauge(la, name, fused):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    T, X, N = int(g["T"]), int(g["X"]), int(g["num_steps"])
    arch = str(g["arch"]) if "arch" in g.files else "generic"
    if "xnet/h_layer/W" in g.files:
        xp = {k[5:]: g[k] for k in g.files if k.startswith("xnet/")}
        vp = {k[5:]: g[k] for k in g.files if k.startswith("vnet/")}
    else:
        mk = H.conv_weights if arch == "conv3D" else H.gauge_weights
        xp, vp = mk(T, X, seed=106, regime=str(g["regime"]))
    B = g["x"].shape[0]
    dyn = H.gauge_hip(T, X, N, float(g["eps"]), xp, vp, g["masks"], B, arch=arch)
    dyn.fused = fused
    beta = float(g["beta"])
    S, Tt, Q = dyn.momentum_fn([g["x"], g["grad0"], dyn._format_time(0)])
    assert H.relerr(np_(S), g["stq0_S"]) < TOL_OP and H.relerr(np_(Tt), g["stq0_T"]) < TOL_OP
    assert H.relerr(np_(Q), g["stq0_Q"]) < TOL_OP
    # step-by-step trace of the forward trajectory against the committed fp64 trace; the fp32 oracle (same
    # weights, run here) measures what the reference's own precision costs at every step
    orc32 = H.gauge_oracle(T, X, N, float(g["eps"]), xp, vp, arch=arch, dtype=np.float32)
    orc32.mask = g["masks"].astype(np.float32)
    t32 = []
    orc32.transition_kernel(g["x"].astype(np.float32), beta, g["v0f"].astype(np.float32), forward=True, trace=t32)
    x, v = g["x"], g["v0f"]
    ld = np.zeros(B)
    for step in range(N):
        x, v, dl = dyn._forward_lf(x, v, beta, step)
        ld = ld + np_(dl)
        assert_fp32_equivalent(np_(x), g["traj_f/x_steps"][step], t32[step][0], f"x after step {step}")
        assert_fp32_equivalent(np_(v), g["traj_f/v_steps"][step], t32[step][1], f"v after step {step}")
        assert_fp32_equivalent(ld, g["traj_f/logdet_steps"][step], t32[step][2], f"log-det after step {step}")
    got = dyn.apply_transition(g["x"], beta, momentum_f=g["v0f"], momentum_b=g["v0b"], coin=g["coin"], u=g["u"])
    f32 = orc32.apply_transition(g["x"].astype(np.float32), beta, g["v0f"].astype(np.float32),
                                 g["v0b"].astype(np.float32), g["coin"], g["u"].astype(np.float32))
    assert_fp32_equivalent(np_(got[0]), g["x_prop"], f32[0], "x_prop")
    assert np.abs(np_(got[2]) - g["p_accept"]).max() < max(TOL_P, P_RATIO * np.abs(f32[2] - g["p_accept"]).max())


# ----------------------------------------------------------------- toy targets / generic Dynamics
def _small(la, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    if name.startswith("mog"):
        tgt = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    else:
        tgt = la.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
    nh = int(g["num_nodes"])
    dyn = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=int(g["trajectory_length"]), eps=float(g["eps"]),
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=nh),
                      use_temperature=True)
    dyn.set_masks(g["masks"])
    dyn.XNet.load_state({k[5:]: g[k] for k in g.files if k.startswith("xnet/")})
    dyn.VNet.load_state({k[5:]: g[k] for k in g.files if k.startswith("vnet/")})
    return g, tgt, dyn


@pytest.mark.parametrize("name", ["mog_cfg2", "scg_cfg1"])
def test_generic_dynamics_and_propose(la, name):
    g, tgt, dyn = _small(la, name)
    x = g["x"]
    assert H.relerr(np_(dyn.energy(x)), g["energy"]) < 2e-5       # SCG: ill-conditioned precision (cond 1e3)
    assert H.relerr(np_(dyn.grad_energy(x)), g["grad_energy"]) < TOL_OP
    Xf, Vf, pf = dyn.forward(x, init_v=g["v0f"])
    Xb, Vb, pb = dyn.backward(x, init_v=g["v0b"])
    for got, key in ((Xf, "Xf"), (Vf, "Vf"), (Xb, "Xb"), (Vb, "Vb")):
        assert H.relerr(np_(got), g[key]) < TOL_OP, key
    assert np.abs(np_(pf) - g["pf"]).max() < TOL_P and np.abs(np_(pb) - g["pb"]).max() < TOL_P
    Lx, Lv, px, outs = la.propose(x, dyn, init_v=g["v0f"], do_mh_step=True, init_v_backward=g["v0b"],
                                  dir_bits=g["dir_bits"], u=g["u"])
    assert H.relerr(np_(Lx), g["Lx"]) < TOL_OP and H.relerr(np_(Lv), g["Lv_mixed"]) < TOL_OP
    assert np.abs(np_(px) - g["px"]).max() < TOL_P
    safe = np.abs(g["px"] - g["u"]) > 1e-4
    assert H.relerr(np_(outs[0])[safe], g["x_accept"][safe]) < TOL_OP
    # init_v=None -> Lv is None (sampler.py:43-45), momenta drawn on the device
    Lx2, Lv2, px2, outs2 = la.propose(x, dyn, do_mh_step=False)
    assert Lv2 is None and outs2 == [] and torch.all((px2 >= 0) & (px2 <= 1))
    # log_jac=True returns the log-Jacobian instead of p
    X3, V3, lj = dyn.forward(x, init_v=g["v0f"], log_jac=True)
    assert torch.equal(X3, Xf)
    # temperature divides the energy (utils/dynamics.py:227-236)
    dyn.temperature = 2.0
    assert H.relerr(np_(dyn.energy(x)), g["energy"] / 2.0) < 2e-5
    # an arbitrary callable is accepted (utils/dynamics.py:35-43) and runs layer by layer; a non-callable is not
    assert la.Dynamics(2, lambda x: x.sum(1), trajectory_length=3, eps=0.1, net_factory=la.network).layered
    with pytest.raises(TypeError):
        la.Dynamics(2, 3.0, trajectory_length=3, eps=0.1, net_factory=la.network)


@pytest.mark.parametrize("name,B", [("mog_cfg2", 4096), ("scg_cfg1", 61)])
def test_one_launch_propose_equals_the_piecewise_path_bit_for_bit(la, name, B):
    """l2hmc_small_propose (direction bit, both momenta, both trajectories, mix and MH in ONE launch) against the
    same library streams written out with l2hmc_fill_* and fed to the piecewise path (two-direction trajectory
    launch + l2hmc_mix_accept): identical bits, ragged batch included (61 chains: partial wave)."""
    import ctypes as C
    from l2hmc_amd import _lib
    g, tgt, dyn = _small(la, name)
    rng = np.random.default_rng(5)
    x = torch.as_tensor(rng.standard_normal((B, 2)) * 0.7 + 0.3, dtype=torch.float32, device="cuda")
    seed, d0 = dyn._seed, 10
    L = _lib.lib()

    def fill(fn, n, off):
        out = torch.empty(n, device="cuda")
        _lib.check(fn(out.data_ptr(), n, seed, off, None))
        return out
    bits = (fill(L.l2hmc_fill_uniform, B, d0) >= 0.5).float()
    vf = fill(L.l2hmc_fill_normal, 2 * B, d0 + 1).reshape(B, 2)
    vb = fill(L.l2hmc_fill_normal, 2 * B, d0 + 2).reshape(B, 2)
    u = fill(L.l2hmc_fill_uniform, B, d0 + 3)
    Lx, Lv, px, outs = la.propose(x, dyn, init_v=vf, do_mh_step=True, init_v_backward=vb, dir_bits=bits, u=u)
    # the host class with library draws: one launch, the draw counter moves by four (three without the MH step)
    dyn._draws = d0
    Lx1, Lv1, px1, outs1 = la.propose(x, dyn, do_mh_step=True)
    assert dyn._draws == d0 + 4 and Lv1 is None
    assert torch.equal(Lx1, Lx) and torch.equal(px1, px) and torch.equal(outs1[0], outs[0])
    assert 0.2 < float(bits.mean()) < 0.8 and not torch.equal(outs[0], Lx) and not torch.equal(outs[0], x)
    dyn._draws = d0
    Lx2, _, px2, outs2 = la.propose(x, dyn, do_mh_step=False)
    assert dyn._draws == d0 + 3 and outs2 == [] and torch.equal(Lx2, Lx) and torch.equal(px2, px)
    # the C entry also hands out the mixed momentum
    plan = dyn._plan()
    Lv3 = torch.empty_like(x)
    _lib.check(L.l2hmc_small_propose(C.byref(plan), x.data_ptr(), B, seed, d0, None, Lv3.data_ptr(), None, None, None))
    assert torch.equal(Lv3, Lv)
    hm = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=3, eps=0.1, hmc=True)
    hplan = hm._plan()
    assert L.l2hmc_small_propose(C.byref(hplan), x.data_ptr(), B, seed, d0, None, Lv3.data_ptr(), None, None, None) != 0


@pytest.mark.parametrize("name", ["mog_cfg2", "scg_cfg1"])
def test_layer_by_layer_generic_dynamics_equals_the_one_launch_kernels(la, name):
    """`Dynamics.layered` (l2hmc_amd/dynamics.py: one S/T/Q evaluation through l2hmc_stq_dense and one
    l2hmc_lf_update_v / _x per sub-update of utils/dynamics.py:120-225) is the path for everything the one-launch toy
    kernels do not hold.  On the shapes both paths take -- the committed cfg-1 / cfg-2 fixtures -- it must reproduce
    the fixtures (float64 oracle) at the same bars, and therefore the one-launch kernels."""
    g, tgt, dyn = _small(la, name)
    assert not dyn.layered
    dyn.layered = True
    x = g["x"]
    Xf, Vf, pf = dyn.forward(x, init_v=g["v0f"])
    Xb, Vb, pb = dyn.backward(x, init_v=g["v0b"])
    for got, key in ((Xf, "Xf"), (Vf, "Vf"), (Xb, "Xb"), (Vb, "Vb")):
        assert H.relerr(np_(got), g[key]) < TOL_OP, key
    assert np.abs(np_(pf) - g["pf"]).max() < TOL_P and np.abs(np_(pb) - g["pb"]).max() < TOL_P
    Lx, Lv, px, outs = la.propose(x, dyn, init_v=g["v0f"], do_mh_step=True, init_v_backward=g["v0b"],
                                  dir_bits=g["dir_bits"], u=g["u"])
    assert H.relerr(np_(Lx), g["Lx"]) < TOL_OP and H.relerr(np_(Lv), g["Lv_mixed"]) < TOL_OP
    assert np.abs(np_(px) - g["px"]).max() < TOL_P
    # library draws: propose runs (no one-launch kernel for this instance) and accepts / rejects row by row
    Lx2, Lv2, px2, (out2,) = la.propose(x, dyn, do_mh_step=True)
    assert Lv2 is None and bool(((out2 == Lx2).all(dim=1) | (out2 == _lib_dev(x)).all(dim=1)).all())
    assert float(px2.min()) >= 0.0 and float(px2.max()) <= 1.0


def _lib_dev(a):
    from l2hmc_amd import _lib
    return _lib.as_dev(a)


def test_torch_ops_dispatch_to_the_same_kernels(la):
    """torch.ops.l2hmc.* (l2hmc_amd/torch_ops.py) against the class surface: same library entries, same bits."""
    import l2hmc_amd.torch_ops  # noqa: F401 -- registers the operators
    from l2hmc_amd import _lib
    ops = torch.ops.l2hmc
    rng = np.random.default_rng(17)
    T = X = 8
    B, D = 96, 128
    x = torch.as_tensor(rng.uniform(0, 2 * np.pi, (B, D)), dtype=torch.float32, device="cuda")
    v = torch.as_tensor(rng.standard_normal((B, D)), dtype=torch.float32, device="cuda")
    want = la.u1_observables(x, T, X, beta=2.0, want_force=True)
    action, force, plaq, charge = ops.u1_action_force(x, T, X, 2.0)
    assert torch.equal(action, want["action"]) and torch.equal(force, want["force"])
    assert torch.equal(plaq, want["avg_plaq"]) and torch.equal(charge, want["top_charge"])
    xp, _ = H.gauge_weights(T, X, regime="mild")
    net = la.GenericNet(model_name='XNet', x_dim=D, num_hidden=4 * D, factor=2., name_scope='position', links_shape=(T, X, 2))
    net.load_state(xp)
    tt = np.array([[0.3, 0.95]])
    S0, T0, Q0 = net([v, x, tt])
    S, Tr, Q = ops.stq_dense(v, x, net.flat_tensors(), net.q_tanh, float(np.float32(0.3)), float(np.float32(0.95)))
    assert torch.equal(S, S0) and torch.equal(Tr, T0) and torch.equal(Q, Q0)
    keep = torch.as_tensor((rng.uniform(size=D) < 0.5).astype(np.float32), device="cuda")
    for d in (0, 1):
        v1, ld = ops.lf_update_v(v, force, S, Tr, Q, 0.1, d)
        x1, ld2 = ops.lf_update_x(x, v1, keep, S, Tr, Q, 0.1, d)
        ref_v, ref_ld = torch.empty_like(v), torch.empty(B, device="cuda")
        _lib.check(_lib.lib().l2hmc_lf_update_v(v.data_ptr(), force.data_ptr(), S.data_ptr(), Tr.data_ptr(), Q.data_ptr(),
                                                0.1, d, B, D, ref_v.data_ptr(), ref_ld.data_ptr(), _lib.stream_ptr()))
        assert torch.equal(v1, ref_v) and torch.equal(ld, ref_ld)
        assert torch.equal(x1[:, keep > 0.5], x[:, keep > 0.5]) and bool(torch.isfinite(ld2).all())
    k = ops.kinetic_energy(v)
    assert H.relerr(np_(k), 0.5 * (np_(v).astype(np.float64) ** 2).sum(1)) < 1e-6
    p = ops.accept_prob(k, k + 0.5, torch.zeros_like(k))
    assert H.relerr(np_(p), np.full(B, np.exp(-0.5))) < 1e-6
    y = ops.wrap_angle(x + 7.0)
    assert float(y.min()) >= 0.0 and float(y.max()) < 2 * np.pi + 1e-6
    coin = torch.as_tensor(rng.uniform(size=B), dtype=torch.float32, device="cuda")
    u = torch.as_tensor(rng.uniform(size=B), dtype=torch.float32, device="cuda")
    xpz, vpz, pz, xo = ops.mix_accept(x, x1, v1, p, x, v, p * 0.5, coin, u, 1)
    fwd = coin > 0.5
    assert torch.equal(xpz[fwd], x1[fwd]) and torch.equal(xpz[~fwd], x[~fwd])
    acc = pz > u
    assert torch.equal(xo[acc], xpz[acc]) and torch.equal(xo[~acc], x[~acc])


class _QuarticTarget:
    """An energy the packed targets cannot express: E(x) = sum_d (x_d^2 - 1)^2 / 4 + c sum_d x_d x_{d+1} (periodic)."""

    def __init__(self, c):
        self.c = c

    def energy(self, x):
        return np.sum((x * x - 1.) ** 2, axis=1) / 4. + self.c * np.sum(x * np.roll(x, -1, axis=1), axis=1)

    def grad_energy(self, x):
        return (x * x - 1.) * x + self.c * (np.roll(x, -1, axis=1) + np.roll(x, 1, axis=1))


@pytest.mark.parametrize("x_dim,nodes,kind", [(12, 100, "gmm"), (20, 128, "callable"), (3, 70, "callable"), (9, 16, "gmm")])
def test_generic_dynamics_any_energy_any_width(la, x_dim, nodes, kind):
    """The reference's `Dynamics` takes ANY `energy_function`, x_dim and `net_factory` (utils/dynamics.py:35-43,
    utils/network.py:89).  Shapes beyond the one-launch kernels (x_dim > 8, more than 64 hidden units) and an energy
    given as a plain torch callable (gradient by torch.autograd, the reference's tf.gradients) run layer by layer:
    forward, backward, accept probabilities, the inverse pair and `propose` against the float64 oracle."""
    from oracle import dynamics as ogen
    rng = np.random.default_rng(41 + x_dim)
    N, eps, B = 4, 0.1, 96
    if kind == "gmm":
        K = 3
        mus = [rng.normal(0, 1.0, x_dim) for _ in range(K)]
        sig = [np.diag(rng.uniform(0.3, 0.8, x_dim)) for _ in range(K)]
        pis = [0.5, 0.3, 0.2]
        otgt = ogen.GMM(mus, sig, pis)
        if x_dim <= 8:
            fn = la.GMM(mus, sig, pis).get_energy_function()
        else:                         # beyond the packed targets: the same mixture as a torch function
            mu_t = torch.tensor(np.stack(mus), dtype=torch.float32, device="cuda")
            prec_t = torch.tensor(np.stack([np.diag(1. / np.diag(s)) for s in sig]), dtype=torch.float32, device="cuda")
            lc_t = torch.tensor([np.log(p / np.sqrt((2 * np.pi) ** x_dim * np.linalg.det(s))) for p, s in zip(pis, sig)],
                                dtype=torch.float32, device="cuda")

            def fn(x):
                d = x[:, None, :] - mu_t[None]
                q = -0.5 * torch.einsum("bkd,kde,bke->bk", d, prec_t, d) + lc_t[None]
                return -torch.logsumexp(q, dim=1)
    else:
        otgt = _QuarticTarget(0.3)

        def fn(x):
            return ((x * x - 1.) ** 2).sum(dim=1) / 4. + 0.3 * (x * torch.roll(x, -1, dims=1)).sum(dim=1)
    xp, vp = H.mlp_weights(x_dim, nodes, seed=7, regime="stress")
    masks = ogen.make_masks(N, x_dim, np.random.RandomState(3))
    orc = ogen.DynamicsOracle(x_dim, otgt, N, eps, masks, xp, vp)
    dyn = la.Dynamics(x_dim, fn, trajectory_length=N, eps=eps,
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=nodes))
    assert dyn.layered
    dyn.set_masks(masks)
    dyn.XNet.load_state(xp)
    dyn.VNet.load_state(vp)
    x = rng.normal(0, 0.8, (B, x_dim))
    v0 = rng.standard_normal((B, x_dim))
    assert H.relerr(np_(dyn.energy(x)), orc.energy(x)) < 2e-5
    assert H.relerr(np_(dyn.grad_energy(x)), orc.grad_energy(x)) < TOL_OP
    Xf, Vf, pf = dyn.forward(x, init_v=v0)
    Xb, Vb, pb = dyn.backward(x, init_v=v0)
    wf, wb = orc.forward(x, v0), orc.backward(x, v0)
    for got, want in ((Xf, wf[0]), (Vf, wf[1]), (Xb, wb[0]), (Vb, wb[1])):
        assert H.relerr(np_(got), want) < 2 * TOL_OP
    assert np.abs(np_(pf) - wf[2]).max() < TOL_P and np.abs(np_(pb) - wb[2]).max() < TOL_P
    assert float(pf.mean()) > 0.01
    # backward undoes forward and the log-determinants cancel (utils/dynamics.py:172-225)
    Xr, Vr, ljb = dyn.backward(Xf, init_v=Vf, log_jac=True)
    _, _, ljf = dyn.forward(x, init_v=v0, log_jac=True)
    assert H.relerr(np_(Xr), x) < 1e-4 and H.relerr(np_(Vr), v0) < 1e-4
    assert float((ljf + ljb).abs().max()) < 1e-4 * max(1.0, float(ljf.abs().max()))
    dir_bits, u = rng.integers(0, 2, B).astype(np.float32), rng.uniform(size=B)
    Lx, Lv, px, (out,) = la.propose(x, dyn, init_v=v0, do_mh_step=True, init_v_backward=v0, dir_bits=dir_bits, u=u)
    want = ogen.propose(x, orc, v0, v0, dir_bits, u=u, do_mh_step=True)
    assert H.relerr(np_(Lx), want[0]) < 2 * TOL_OP and np.abs(np_(px) - want[2]).max() < TOL_P


@pytest.mark.parametrize("name,B", [("mog_cfg2", 4096), ("mog_cfg2", 37), ("scg_cfg1", 128)])
def test_both_first_layer_forms_of_the_toy_kernel_agree(la, name, B):
    """Batches of at most one wave per SIMD evaluate the toy networks' first layer on the matrix pipe, chip-filling
    ones on the VALU (csrc/small_mlp.hip); the forms walk the hidden layer's k in different orders, so `propose` and
    both trajectory directions must agree to fp32 rounding (TOL_OP / TOL_P), and each form is deterministic."""
    from l2hmc_amd import _lib
    g, tgt, dyn = _small(la, name)
    rng = np.random.default_rng(11)
    x = torch.as_tensor(rng.standard_normal((B, 2)) * 0.7 + 0.3, dtype=torch.float32, device="cuda")
    v = torch.as_tensor(rng.standard_normal((B, 2)), dtype=torch.float32, device="cuda")
    outs = {}
    try:
        # 1 = first layer on the matrix pipe, 2 = on the VALU, 3 = two waves per group (l2hmc_small_plan::first_layer_form)
        for form in ((1, 2, 3, 1, 3) if name.startswith("mog") else (1, 2, 1)):
            dyn.first_layer_form = form
            dyn._draws = 20
            Lx, _, px, mh = la.propose(x, dyn, do_mh_step=False)
            Xf, Vf, pf = dyn.forward(x, init_v=v)
            Xb, Vb, pb = dyn.backward(x, init_v=v)
            now = (Lx, px, Xf, Vf, pf, Xb, Vb, pb)
            if form in outs:
                assert all(torch.equal(a, b) for a, b in zip(now, outs[form]))
            outs[form] = now
    finally:
        dyn.first_layer_form = 0
    errs = [H.relerr(np_(a), np_(b)) for a, b in zip(outs[1], outs[2])]
    print("toy kernel, matrix-pipe vs VALU first layer:", " ".join(f"{e:.1e}" for e in errs))
    # two fp32 summation orders of the same N-step trajectory (measured: profiles/r03_gate_envelope.txt)
    assert max(errs) < FORMS_TOL, errs
    if 3 in outs:
        errs3 = [H.relerr(np_(a), np_(b)) for a, b in zip(outs[1], outs[3])]
        print("toy kernel, one wave vs two waves per group:", " ".join(f"{e:.1e}" for e in errs3))
        assert max(errs3) < FORMS_TOL, errs3
    assert float(outs[1][1].mean()) > 0.01          # not a trivially rejected batch


def test_generic_dynamics_hmc_and_quadratic_gaussian(la):
    tgt = la.Gaussian(np.zeros(2), np.array([[1.0, 0.3], [0.3, 0.5]]))
    ot = ogen.Gaussian(np.zeros(2), np.array([[1.0, 0.3], [0.3, 0.5]]))
    dyn = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=4, eps=0.05, hmc=True)
    rng = np.random.default_rng(3)
    x, v = rng.standard_normal((50, 2)), rng.standard_normal((50, 2))
    orc = ogen.DynamicsOracle(2, ot, 4, 0.05, np_(dyn.mask), hmc=True)
    X, V, p = dyn.forward(x, init_v=v)
    want = orc.forward(x, v)
    assert H.relerr(np_(X), want[0]) < TOL_OP and H.relerr(np_(V), want[1]) < TOL_OP
    assert np.abs(np_(p) - want[2]).max() < TOL_P
    q = la.quadratic_gaussian(x, np.zeros(2), ot.i_sigma)
    assert H.relerr(np_(q), ogen.quadratic_gaussian(x, np.zeros(2), ot.i_sigma.astype('float32').astype(float))) < TOL_OP


# ----------------------------------------------------------------- RNG
def _philox_ref(block, offset, seed):
    """Philox4x32-10 in Python integers (Salmon et al., SC'11), for the bit-exact check."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [block & 0xFFFFFFFF, block >> 32, offset & 0xFFFFFFFF, offset >> 32]
    k0, k1 = seed & 0xFFFFFFFF, seed >> 32
    for _ in range(10):
        p0, p1 = c[0] * M0, c[2] * M1
        c = [(p1 >> 32) ^ c[1] ^ k0, p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k1, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def test_philox_stream_bit_exact_and_statistics(la):
    from l2hmc_amd import _lib
    # Random123 known-answer vector: counter = key = 0
    assert _philox_ref(0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    n = 4096 + 3                         # ragged tail
    u = torch.empty(n, device="cuda")
    for seed, off in ((0, 0), (12345678901234, 77)):
        _lib.check(_lib.lib().l2hmc_fill_uniform(u.data_ptr(), n, seed, off, None))
        got = u.cpu().numpy()
        for i in (0, 1, 2, 3, 4, 1001, n - 1):
            w = _philox_ref(i >> 2, off, seed)[i & 3]
            assert got[i] == np.float32((w >> 8) * 2.0 ** -24)      # integer work: bit-exact
    z = torch.empty(1 << 20, device="cuda")
    _lib.check(_lib.lib().l2hmc_fill_normal(z.data_ptr(), z.numel(), 7, 0, None))
    zz = z.double().cpu().numpy()
    assert abs(zz.mean()) < 5e-3 and abs(zz.std() - 1) < 5e-3 and np.isfinite(zz).all()
    assert abs((zz ** 4).mean() - 3.0) < 0.05 and abs(np.corrcoef(zz[::2], zz[1::2])[0, 1]) < 5e-3
    z2 = torch.empty_like(z)
    _lib.check(_lib.lib().l2hmc_fill_normal(z2.data_ptr(), z2.numel(), 7, 0, None))
    assert torch.equal(z, z2)            # same (seed, offset) -> same stream
    _lib.check(_lib.lib().l2hmc_fill_normal(z2.data_ptr(), z2.numel(), 7, 1, None))
    assert not torch.equal(z, z2)


# ----------------------------------------------------------------- sampler-level checks
def test_accept_rate_parity_on_identical_inputs(la):
    """north_star: accept-rate parity +-1 %.  A 60-step chain on the device; at every step the oracle sees the
    same state and the same draws (chains would otherwise diverge chaotically after a few accept flips)."""
    T = X = 8
    N, eps, beta, B, steps = 5, 0.08, 2.0, 32, 60
    orc, _, dyn = _pair(T, X, N, eps, B, "init")
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 2 * np.pi, (B, 128)).astype(np.float32)
    p_hip, p_orc, acc_hip, acc_orc = [], [], [], []
    for _ in range(steps):
        v0f, v0b = rng.standard_normal((B, 128)), rng.standard_normal((B, 128))
        coin, u = rng.uniform(size=B), rng.uniform(size=B)
        got = dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
        want = orc.apply_transition(x.astype(np.float64), beta, v0f, v0b, coin, u)
        p_hip.append(np_(got[2]))
        p_orc.append(want[2])
        acc_hip.append(np_(got[2]) > u)
        acc_orc.append(want[2] > u)
        x = np.mod(got[3].cpu().numpy(), 2 * np.pi).astype(np.float32)     # gauge_model.py:1180
    p_hip, p_orc = np.concatenate(p_hip), np.concatenate(p_orc)
    assert 0.02 < p_orc.mean() < 0.98                      # a non-degenerate acceptance regime
    assert abs(p_hip.mean() - p_orc.mean()) < 1e-4         # bar: 1e-2
    assert abs(np.mean(acc_hip) - np.mean(acc_orc)) < 2e-3
    assert np.abs(p_hip - p_orc).max() < 1e-4


def test_hmc_sampling_reproduces_exact_plaquette(la):
    """The reference prints I1(beta)/I0(beta) beside the measured plaquette at every step
    (lattice.py:31-33, gauge_model.py:1149,1216).  Plain-HMC mode (gauge_dynamics.py:102-108) through the
    whole device-side MCMC step -- Philox draws, both directions, mixing, MH -- must reproduce it."""
    beta, L, B = 2.0, 8, 1024
    torch.manual_seed(0)
    lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=B, rand=False)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.12, hmc=True, num_steps=8, eps_trainable=False,
                           network_arch='generic', seed=5)
    x = torch.rand(B, 2 * L * L, device="cuda") * (2 * np.pi)
    plaqs, accs = [], []
    for step in range(140):
        _, _, p, x_out = dyn(x, beta)
        x = torch.remainder(x_out, 2 * np.pi)
        if step >= 40:
            plaqs.append(float(la.u1_observables(x, L, L)["avg_plaq"].mean()))
            accs.append(float(p.mean()))
    exact = la.u1_plaq_exact(beta)
    assert 0.5 < np.mean(accs) < 1.0
    assert abs(np.mean(plaqs) - exact) < 4e-3, (np.mean(plaqs), exact)


@pytest.mark.parametrize("L,arch,B", [(8, "generic", 70), (4, "generic", 9), (8, "conv3D", 21), (16, "generic", 5)])
def test_native_mcmc_step_matches_oracle_on_its_own_draws(la, L, arch, B):
    """l2hmc_gauge_mcmc_step (draws + both trajectories + mix/MH + observables + wrap in one call).  Its Philox
    streams are reproducible through l2hmc_fill_*, so the oracle can be fed the very same draws."""
    import ctypes as C
    from l2hmc_amd import _lib
    N, eps, beta, D = 3, 0.1, 2.0, 2 * L * L
    xp, vp = (H.conv_weights if arch == "conv3D" else H.gauge_weights)(L, L, regime="mild")
    orc = H.gauge_oracle(L, L, N, eps, xp, vp, arch=arch)
    dyn = H.gauge_hip(L, L, N, eps, xp, vp, orc.mask, B, arch=arch)
    x0 = np.random.default_rng(3).uniform(0, 2 * np.pi, (B, D)).astype(np.float32)
    x = torch.as_tensor(x0, device="cuda").clone()
    outs = [torch.empty(B, device="cuda") for _ in range(5)]
    plan, Lh = dyn._plan(), _lib.lib()
    ws, nb = dyn._ws.get(Lh.l2hmc_gauge_mcmc_step_ws_bytes(C.byref(plan), B), x.device)
    seed, draw = 77, 5
    _lib.check(Lh.l2hmc_gauge_mcmc_step(C.byref(plan), beta, x.data_ptr(), B, seed, draw, *[o.data_ptr() for o in outs],
                                         ws, nb, _lib.stream_ptr()))
    V = torch.empty(2 * B, D, device="cuda")
    cu = torch.empty(2 * B, device="cuda")
    _lib.check(Lh.l2hmc_fill_normal(V.data_ptr(), V.numel(), seed, 2 * draw, None))
    _lib.check(Lh.l2hmc_fill_uniform(cu.data_ptr(), cu.numel(), seed, 2 * draw + 1, None))
    V, cu = np_(V), np_(cu)
    x64 = x0.astype(np.float64)
    want = orc.apply_transition(x64, beta, V[:B], V[B:], cu[:B], cu[B:])
    px, actions, plaqs, charges, dq = [np_(o) for o in outs]
    assert np.abs(px - want[2]).max() < TOL_P
    assert H.relerr(actions, olat.total_action(x64, L, L)) < TOL_OP
    assert H.relerr(plaqs, olat.avg_plaq(x64, L, L)) < TOL_OP
    assert H.relerr(charges, olat.top_charge(x64, L, L)) < 1e-4
    safe = np.abs(want[2] - cu[B:]) > 1e-4
    assert H.relerr(dq[safe], olat.top_charge_diff(x64, want[3], L, L)[safe]) < 1e-3
    xw = np.mod(want[3], 2 * np.pi)
    got = np_(x)
    d = np.abs(got - xw)[safe]
    d = np.minimum(d, 2 * np.pi - d)           # a value a hair below 0 wraps to just under 2 pi
    assert d.max() < 5e-5
    assert got.min() >= 0 and got.max() < 2 * np.pi + 1e-6


@pytest.mark.parametrize("metric", ["cos_diff", "l2", "cos"])
def test_loss_forward_matches_oracle(la, metric):
    """gauge_model.py:728-797 forward value; the kernel gets the oracle's own proposals so only the loss
    arithmetic is under test, then the end-to-end method is checked for consistency."""
    from oracle import loss as oloss
    from l2hmc_amd import _lib
    T = X = 8
    B = 24
    rng = np.random.default_rng(4)
    x, xp, z = (rng.uniform(0, 2 * np.pi, (B, 128)) for _ in range(3))
    px, pz = rng.uniform(0.05, 1, B), rng.uniform(0.05, 1, B)
    want = oloss.calc_loss_terms(x, xp, px, z, pz, T, X, metric=metric, loss_scale=0.7, aux_weight=0.9, std_weight=1.1,
                                 charge_weight=1.3)
    dev = [torch.as_tensor(a, dtype=torch.float32, device="cuda").contiguous() for a in (x, xp, px, z, pz)]
    terms = torch.empty(B, device="cuda")
    _lib.check(_lib.lib().l2hmc_gauge_loss_terms(*[d.data_ptr() for d in dev], B, T, X,
                                                 la.GaugeSampler.METRICS[metric], 0.7, 0.9, 1.1, 1.3, terms.data_ptr(),
                                                 None))
    assert np.max(np.abs(np_(terms) - want) / np.maximum(1.0, np.abs(want))) < 5e-5
    if metric == "cos_diff":
        orc, _, dyn = _pair(T, X, 3, 0.1, B, "mild")
        smp = la.GaugeSampler(dyn)
        loss, x_out, pxd, x_dq = smp.calc_loss(x, 2.0, metric=metric, z=z)
        assert abs(float(loss) - float(smp.last_loss_terms.mean())) < 1e-4 * max(1.0, abs(float(loss)))
        assert x_out.shape == (B, 128) and x_dq.dtype == torch.int32 and torch.isfinite(loss)
        with pytest.raises(AttributeError):
            smp.calc_loss(x, 2.0, metric="bogus")


def test_device_resident_sampling_loop(la):
    """GaugeSampler.run (gauge_model.py:1304-1460 without files/plots): wrap on the device equals np.mod,
    histories have the reference's shapes, beta annealing follows :1039-1046."""
    T = X = 8
    B = 48
    orc, _, dyn = _pair(T, X, 3, 0.1, B, "mild")
    smp = la.GaugeSampler(dyn, beta_init=2., beta_final=4., train_steps=100)
    assert abs(smp.update_beta(0) - 2.0) < 1e-12 and abs(smp.update_beta(100) - 4.0) < 1e-12
    assert abs(1. / smp.update_beta(50) - 0.5 * (1 / 2. + 1 / 4.)) < 1e-12
    xs = torch.randn(B, 128, device="cuda") * 20.
    w = smp.wrap(xs)
    np.testing.assert_allclose(np_(w), np.mod(xs.cpu().numpy(), np.float32(2 * np.pi)), atol=2e-6)
    assert float(w.min()) >= 0 and float(w.max()) < 2 * np.pi + 1e-6
    # selected-only mode (L2HMC_PLAN_SELECTED_ONLY) must give the same kind of step
    dyn.both_directions = False
    xa, pxa, obsa, dqa = smp.step(w.clone(), 2.0)
    assert xa.shape == w.shape and float(xa.min()) >= 0 and torch.all((pxa >= 0) & (pxa <= 1))
    assert H.relerr(np_(obsa["action"]), olat.total_action(np_(w), T, X)) < TOL_OP
    dyn.both_directions = True
    out = smp.run(5, 2.0, keep_samples=True)
    assert out["px"].shape == (5, B) and out["samples"].shape == (5, B, 128)
    assert np.all((out["px"] >= 0) & (out["px"] <= 1)) and abs(out["plaq_exact"] - 0.697775) < 1e-6
    # observables of step k are those of step k-1's output samples
    S = olat.total_action(out["samples"][2].astype(np.float64), T, X)
    assert H.relerr(out["actions"][3], S) < TOL_OP
    q = olat.top_charge(out["samples"][2].astype(np.float64), T, X)
    assert H.relerr(out["charges"][3], q) < 1e-4
    ess = la.stats.ESS(la.stats.acl_spectrum(out["samples"], 1.0) / la.stats.autocovariance(out["samples"], 0))
    assert 0 < ess <= 1.0


# ----------------------------------------------------------------- full size (BASELINE.json configs[2])
def test_full_size_properties_cfg3(la):
    """B=2048, 10 LF steps: size-independent properties instead of an oracle run."""
    T = X = 8
    N, eps, beta, B = 10, 0.25, 2.0, 2048
    xp, vp = H.gauge_weights(T, X, regime="mild")
    masks = H.gauge_oracle(T, X, N, eps, xp, vp).mask
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, masks, B)
    torch.manual_seed(1)
    x = torch.rand(B, 128, device="cuda") * (2 * np.pi)
    v = torch.randn(B, 128, device="cuda")
    # (1) reversibility: backward trajectory undoes the forward one, log-dets cancel
    x1, v1, p1, ld1 = dyn.transition_kernel(x, beta, forward=True, momentum=v, return_logdet=True)
    x2, v2, p2, ld2 = dyn.transition_kernel(x1, beta, forward=False, momentum=v1, return_logdet=True)
    # (20 chaotic steps there and back: rounding noise is amplified twice; 1e-3 still pins the inverse)
    assert H.relerr(np_(x2), np_(x)) < 1e-3 and H.relerr(np_(v2), np_(v)) < 1e-3
    assert H.relerr(np_(ld2), -np_(ld1)) < 1e-3
    # (2) determinism + fused == layered on identical inputs
    x1b, v1b, p1b, _ = dyn.transition_kernel(x, beta, forward=True, momentum=v, return_logdet=True)
    assert torch.equal(x1, x1b) and torch.equal(p1, p1b)
    dyn.fused = False
    x1c, v1c, p1c, ld1c = dyn.transition_kernel(x, beta, forward=True, momentum=v, return_logdet=True)
    # two fp32 summation orders of the same 10 mildly chaotic steps over 2048 chains (measured on the box:
    # profiles/r03_gate_envelope.txt)
    dx_fl, dp_fl = H.relerr(np_(x1c), np_(x1)), np.abs(np_(p1c) - np_(p1)).max()
    print(f"cfg3 full size, fused vs layered: x {dx_fl:.2e}  p {dp_fl:.2e}")
    assert dx_fl < FUSED_VS_LAYERED_X and dp_fl < FUSED_VS_LAYERED_P, (dx_fl, dp_fl)
    dyn.fused = True
    # (3) first 64 chains agree with the oracle; p in [0, 1]
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    orc32 = H.gauge_oracle(T, X, N, eps, xp, vp, dtype=np.float32)
    xs, vs = np_(x[:64]), np_(v[:64])
    want = orc.transition_kernel(xs, beta, vs, forward=True)
    w32 = orc32.transition_kernel(xs.astype(np.float32), beta, vs.astype(np.float32), forward=True)
    assert_fp32_equivalent(np_(x1[:64]), want[0], w32[0], "x (first 64 chains)")
    assert torch.all((p1 >= 0) & (p1 <= 1))
    # (4) a whole MCMC step with device-side draws: outputs are either the proposal or the input
    out = dyn(x, beta)
    same = (out[3] == x).all(1) | (out[3] == out[0]).all(1)
    assert bool(same.all())


def test_check_numerics_opt_in(la):
    """gauge_dynamics.py:26-28: the reference aborts a step whose exp overflows; here NaN/inf propagate by default
    (non-finite accept probability -> 0, :609) and `check_numerics=True` turns that into an exception."""
    T = X = 4
    xp, vp = H.gauge_weights(T, X, regime="stress")
    for k in ("scale_layer/b",):
        xp[k] = xp[k] + 400.0          # exp(eps * S) overflows fp32 in the position update
    xp["coeff_scale"] = xp["coeff_scale"] + 6.0
    orc = H.gauge_oracle(T, X, 2, 0.3, xp, vp)
    dyn = H.gauge_hip(T, X, 2, 0.3, xp, vp, orc.mask, 6)
    x, v0f, v0b, coin, u = H.gauge_inputs(6, 32)
    x_prop, v_prop, p, x_out = dyn.apply_transition(x, 2.0, v0f, v0b, coin, u)
    assert not torch.isfinite(x_prop).all()
    assert torch.isfinite(p).all() and float(p.max()) == 0.0           # non-finite accept probability -> 0 (:609)
    # ... but the reference's accept/reject is a*x_prop + (1-a)*x (:252-257), so 0 * NaN still poisons the state
    assert not torch.isfinite(x_out).all()
    dyn.check_numerics = True
    with pytest.raises(FloatingPointError):
        dyn.apply_transition(x, 2.0, v0f, v0b, coin, u)


def test_c_abi_from_plain_c_host_program(tmp_path):
    """The drop-in boundary is a C ABI: a C99 program compiled with gcc (no Python, no torch, no HIP compiler)
    allocates device memory through the HIP runtime and drives the library (examples/c_abi_demo.c)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "c_abi_demo")
    libdir = os.path.join(root, "l2hmc_amd")
    cmd = ["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_abi_demo.c"),
           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-L" + libdir, "-l:libl2hmc_hip.so",
           "-L/opt/rocm/lib", "-lamdhip64", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c_abi_demo OK" in out.stdout and "cold start: action 0.000000  avg_plaq 1.000000" in out.stdout


def test_sampler_run_is_saved_as_npz_and_statistics_text(la, tmp_path):
    """f4: observables + statistics on disk (gauge_model.py:1758-2033), npz and text instead of pickles."""
    T = X = 4
    xp, vp = H.gauge_weights(T, X, regime="init")
    orc = H.gauge_oracle(T, X, 3, 0.1, xp, vp)
    dyn = H.gauge_hip(T, X, 3, 0.1, xp, vp, orc.mask, 16)
    smp = la.GaugeSampler(dyn)
    out = smp.run(40, 2.0, keep_samples=True)
    npz, txt = smp.save_run(out, str(tmp_path), 2.0, therm_frac=4)
    with np.load(npz, allow_pickle=False) as f:
        assert f["plaqs"].shape == (40, 16) and f["samples"].shape == (40, 16, 32) and float(f["beta"]) == 2.0
        np.testing.assert_array_equal(f["px"], out["px"])
    text = open(txt).read()
    assert "average plaquette" in text and "exact 0.697775" in text and "charge probabilities" in text


@pytest.mark.parametrize("arch,fused,B", [("generic", True, 70), ("generic", False, 33), ("conv3D", True, 40)])
def test_selected_only_step_equals_both_directions_step(la, arch, fused, B):
    """L2HMC_PLAN_SELECTED_ONLY: every chain is integrated only in the direction its coin picks, with the momentum
    the both-directions step would have drawn for that direction (same Philox streams) -- the chains must
    coincide: the reference's mix multiplies the unselected trajectory by an exact 0."""
    T = X = 8
    xp, vp = (H.conv_weights if arch == "conv3D" else H.gauge_weights)(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, 4, 0.15, xp, vp, arch=arch)
    dyn = H.gauge_hip(T, X, 4, 0.15, xp, vp, orc.mask, B, arch=arch)
    dyn.fused = fused
    x0 = torch.rand(B, 128, device="cuda") * (2 * np.pi)
    res = {}
    for both in (True, False):
        dyn.both_directions = both
        dyn._draws = 0                                  # replay the same Philox streams in both modes
        smp = la.GaugeSampler(dyn)
        x, hist = x0.clone(), []
        for _ in range(3):
            x, px, obs, dq = smp.step(x, 2.0)
            hist.append((x.clone(), px.clone(), obs["top_charge"].clone(), dq.clone()))
        res[both] = hist
    for a, b in zip(res[True], res[False]):
        for s_, t_ in zip(a, b):
            if fused:
                assert torch.equal(s_, t_)              # per-row arithmetic does not depend on the row's tile mates
            else:
                assert H.relerr(np_(t_), np_(s_)) < 1e-6
    assert 0.2 < float((res[True][0][1] > 0).float().mean()) <= 1.0


@pytest.mark.parametrize("B", [64, 2048, 6200])
def test_mcmc_step_is_hip_graph_capturable(la, B):
    """include/l2hmc_hip.h promises no allocation / synchronisation inside the library: a whole MCMC step
    (draws + trajectories + mix/accept + observables + wrap) is captured into a HIP graph and replayed -- as one
    sub-tile launch (64 chains), one 16-row launch (2048) and the three launches of a cut batch (6200: 32-row rounds,
    a 16-row round, a sub-tile launch; launch_fused_step)."""
    import ctypes as C
    from l2hmc_amd import _lib
    T = X = 8
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, 3, 0.1, xp, vp)
    dyn = H.gauge_hip(T, X, 3, 0.1, xp, vp, orc.mask, B)
    plan, L = dyn._plan(), _lib.lib()
    nb = L.l2hmc_gauge_mcmc_step_ws_bytes(C.byref(plan), B)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    x0 = torch.rand(B, 128, device="cuda") * (2 * np.pi)
    outs = [torch.empty(B, device="cuda") for _ in range(5)]

    def step(x):
        _lib.check(L.l2hmc_gauge_mcmc_step(C.byref(plan), 2.0, x.data_ptr(), B, 42, 7, *(o.data_ptr() for o in outs),
                                           ws.data_ptr(), nb, _lib.stream_ptr()))
    eager = x0.clone()
    step(eager)                                   # also the warm-up (one-time function attributes are set here)
    torch.cuda.synchronize()
    want_px = outs[0].clone()
    xg = x0.clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            step(xg)
    torch.cuda.current_stream().wait_stream(side)
    xg.copy_(x0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(xg, eager) and torch.equal(outs[0], want_px)
    g.replay()                                    # a second replay advances the chains from the first one's output
    torch.cuda.synchronize()
    assert not torch.equal(xg, eager) and float(xg.min()) >= 0.0


def test_reference_quirk_wrap_breaks_torus_reversibility(la):
    """Quirk Q10: the trajectory map is invertible (backward o forward = id), but the reference wraps the
    state to [0, 2 pi) between steps and its networks act on raw angles, so the map does not commute with 2 pi
    shifts: inverting from the WRAPPED image does not come back.  Reproduced, not 'fixed' (drop-in contract)."""
    T = X = 8
    B = 64
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, 5, 0.2, xp, vp)
    dyn = H.gauge_hip(T, X, 5, 0.2, xp, vp, orc.mask, B)
    x = torch.rand(B, 128, device="cuda") * (2 * np.pi)
    v = torch.randn(B, 128, device="cuda")
    x1, v1, _ = dyn.transition_kernel(x, 2.0, forward=True, momentum=v)
    ang = lambda a, b: torch.remainder(a - b + np.pi, 2 * np.pi) - np.pi   # noqa: E731
    x2, _, _ = dyn.transition_kernel(x1, 2.0, forward=False, momentum=v1)
    assert float(ang(x2, x).abs().max()) < 1e-3
    assert float((x1 < 0).float().mean() + (x1 >= 2 * np.pi).float().mean()) > 0.01      # some links left [0, 2 pi)
    x3, _, _ = dyn.transition_kernel(torch.remainder(x1, 2 * np.pi), 2.0, forward=False, momentum=v1)
    assert float(ang(x3, x).pow(2).mean().sqrt()) > 1e-2
    # with no networks (plain HMC) the map is equivariant and the wrap is harmless
    hmc = H.gauge_hip(T, X, 5, 0.2, xp, vp, orc.mask, B, hmc=True)
    y1, w1, _ = hmc.transition_kernel(x, 2.0, forward=True, momentum=v)
    y3, _, _ = hmc.transition_kernel(torch.remainder(y1, 2 * np.pi), 2.0, forward=False, momentum=w1)
    assert float(ang(y3, x).abs().max()) < 1e-3


def test_generic_dynamics_in_three_dimensions(la):
    """x_dim = 3 runs the run-time-dimension instances of the toy-target kernels (x_dim 2 has compile-time ones):
    trajectories, accept probabilities and `propose` against the float64 oracle."""
    from oracle import dynamics as od
    mus = [np.array([1., 0., 0.5]), np.array([0., 1., -0.5]), np.array([-1., -1., 0.])]
    covs = [np.diag([0.05, 0.08, 0.1]), 0.07 * np.eye(3) + 0.02, np.diag([0.1, 0.05, 0.06])]
    pis = [0.3, 0.5, 0.2]
    tgt_o, tgt = od.GMM(mus, covs, pis), la.GMM(mus, covs, pis)
    for nh in (50, 12):
        N, eps, B = 4, 0.1, 33
        xp, vp = H.mlp_weights(3, nh, regime="stress")
        masks = od.make_masks(N, 3, np.random.RandomState(5))
        orc = od.DynamicsOracle(3, tgt_o, N, eps, masks, xp, vp)
        dyn = la.Dynamics(3, tgt.get_energy_function(), trajectory_length=N, eps=eps,
                          net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=nh))
        dyn.set_masks(masks)
        dyn.XNet.load_state(xp)
        dyn.VNet.load_state(vp)
        rng = np.random.default_rng(8)
        x = tgt_o.get_samples(B, rng)
        v0f, v0b = rng.standard_normal((B, 3)), rng.standard_normal((B, 3))
        bits, u = rng.integers(0, 2, B).astype(np.float64), rng.uniform(size=B)
        assert H.relerr(np_(dyn.energy(x)), orc.energy(x.astype(np.float32).astype(np.float64))) < 2e-5
        assert H.relerr(np_(dyn.grad_energy(x)), orc.grad_energy(x.astype(np.float32).astype(np.float64))) < TOL_OP
        Xf, Vf, pf = dyn.forward(x, init_v=v0f)
        Xb, Vb, pb = dyn.backward(x, init_v=v0b)
        oXf, oVf, opf = orc.forward(x, v0f)
        oXb, oVb, opb = orc.backward(x, v0b)
        for got, want in ((Xf, oXf), (Vf, oVf), (Xb, oXb), (Vb, oVb)):
            assert H.relerr(np_(got), want) < TOL_OP
        assert np.abs(np_(pf) - opf).max() < TOL_P and np.abs(np_(pb) - opb).max() < TOL_P
        Lx, Lv, px, outs = la.propose(x, dyn, init_v=v0f, do_mh_step=True, init_v_backward=v0b, dir_bits=bits, u=u)
        want = od.propose(x, orc, v0f, v0b, bits, u=u, do_mh_step=True)
        assert H.relerr(np_(Lx), want[0]) < TOL_OP and np.abs(np_(px) - want[2]).max() < TOL_P
        # the one-launch propose (run-time-dimension instance, three-component target: LDS-resident parameters)
        # against the piecewise path on the same library streams, bit for bit
        from l2hmc_amd import _lib
        Lh, d0 = _lib.lib(), 20

        def fill(fn, n, off):
            out = torch.empty(n, device="cuda")
            _lib.check(fn(out.data_ptr(), n, dyn._seed, off, None))
            return out
        fb = (fill(Lh.l2hmc_fill_uniform, B, d0) >= 0.5).float()
        fvf = fill(Lh.l2hmc_fill_normal, 3 * B, d0 + 1).reshape(B, 3)
        fvb = fill(Lh.l2hmc_fill_normal, 3 * B, d0 + 2).reshape(B, 3)
        fu = fill(Lh.l2hmc_fill_uniform, B, d0 + 3)
        pLx, _, ppx, pouts = la.propose(x, dyn, init_v=fvf, do_mh_step=True, init_v_backward=fvb, dir_bits=fb, u=fu)
        dyn._draws = d0
        oLx, _, opx, oouts = la.propose(x, dyn, do_mh_step=True)
        assert torch.equal(oLx, pLx) and torch.equal(opx, ppx) and torch.equal(oouts[0], pouts[0])


@pytest.mark.parametrize("T,X", [(4, 16), (16, 4), (2, 32)])
def test_fused_kernel_on_non_square_lattices(la, T, X):
    """D = 2*T*X = 128 selects the whole-trajectory kernel for any T x X = 64 (its plaquette stencil indexes sites with
    shifts: X is a power of two); non-square extents against the float64 oracle, and against the layered path."""
    N, eps, beta, B = 3, 0.15, 2.0, 21
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, 128)
    want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    got = {}
    for fused in (True, False):
        dyn.fused = fused
        got[fused] = dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
        assert H.relerr(np_(got[fused][0]), want[0]) < 5 * TOL_OP and H.relerr(np_(got[fused][1]), want[1]) < 5 * TOL_OP
        assert np.abs(np_(got[fused][2]) - want[2]).max() < TOL_P
    assert H.relerr(np_(got[True][0]), np_(got[False][0])) < 5 * TOL_OP
    # and the native step (finish kernel indexes the same lattice)
    smp = la.GaugeSampler(dyn)
    xs = torch.as_tensor(x, dtype=torch.float32, device="cuda")
    xn, px, obs, dq = smp.step(xs, beta)
    assert H.relerr(np_(obs["action"]), olat.total_action(x.astype(np.float32).astype(np.float64), T, X)) < TOL_OP
    assert H.relerr(np_(obs["top_charge"]), olat.top_charge(x.astype(np.float32).astype(np.float64), T, X)) < 1e-4


def test_sampler_and_dynamics_share_one_draw_counter(la):
    """GaugeSampler.step takes its Philox stream pair from dyn._draws (the counter _normal/_uniform
    advance and save_state stores).  After dynamics draws the sampler's momenta are not a replay of them, and two
    samplers on one dynamics never repeat each other's noise."""
    from l2hmc_amd import _lib
    T = X = 8
    B = 32
    orc, _, dyn = _pair(T, X, 2, 0.1, B, "mild")
    x = torch.rand(B, 128, device="cuda") * (2 * np.pi)
    v_first = dyn._normal((2 * B, 128)).clone()           # stream (seed, 0): what a fresh step_count=0 sampler used to replay
    assert dyn._draws == 1
    s1, s2 = la.GaugeSampler(dyn), la.GaugeSampler(dyn)
    a = s1.step(x, 2.0)
    assert dyn._draws == 4                                 # pair (2, 3)
    b = s2.step(x, 2.0)
    assert dyn._draws == 6                                 # pair (4, 5), not (2, 3) again
    assert not torch.equal(a[1], b[1])                     # same x, different noise -> different accept probabilities
    # the pair the first step consumed is stream 2 / 3, not the stream the dynamics already used
    V = torch.empty(2 * B, 128, device="cuda")
    _lib.check(_lib.lib().l2hmc_fill_normal(V.data_ptr(), V.numel(), dyn._seed, 2, None))
    assert not torch.equal(V, v_first)
    want = orc.apply_transition(np_(x), 2.0, np_(V[:B]), np_(V[B:]), *np.split(np_(_fill_u(dyn._seed, 3, 2 * B)), 2))
    assert np.abs(np_(a[1]) - want[2]).max() < 1e-4        # stream identity (another stream differs by O(1)); hot start
    dyn.apply_transition(x, 2.0)                           # no draw injected: one stream pair (6, 7)
    assert dyn._draws == 8
    dyn.apply_transition(x, 2.0, momentum_f=V[:B])         # partly injected: three single-stream draws
    assert dyn._draws == 11
    s1.step(x, 2.0)                                        # next even-aligned pair (12, 13)
    assert dyn._draws == 14


def _fill_u(seed, offset, n):
    from l2hmc_amd import _lib
    out = torch.empty(n, device="cuda")
    _lib.check(_lib.lib().l2hmc_fill_uniform(out.data_ptr(), n, seed, offset, None))
    return out


# ----------------------------------------------------------------- BASELINE.json configs[4] and configs[3] widths
@pytest.fixture(scope="module")
def cfg5():
    """U(1) 32x32, GenericNet D=2048 / H=8192 (gauge_dynamics.py:169-187: num_hidden = 4 x_dim), beta 4, eps 0.1,
    25 LF steps.  151 M parameters per network: built once for the tests below."""
    T = X = 32
    xp, vp = H.gauge_weights(T, X, regime="mild")
    xp32 = {k: v.astype(np.float32).astype(np.float64) for k, v in xp.items()}   # the values the device holds
    vp32 = {k: v.astype(np.float32).astype(np.float64) for k, v in vp.items()}
    del xp, vp
    return T, X, xp32, vp32


def test_cfg5_leapfrog_steps_at_full_width(la, cfg5):
    """`_forward_lf` / `_backward_lf` (gauge_dynamics.py:412-483) at step 0 and N-1 of the 25-step schedule."""
    T, X, xp, vp = cfg5
    N, eps, beta, B, D = 25, 0.1, 4.0, 6, 2 * T * X
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    assert dyn.position_fn.num_hidden == 8192 and dyn.x_dim == 2048
    x, v, _, _, _ = H.gauge_inputs(B, D, seed=105)
    x, v = x.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64)
    for step in (0, N - 1):
        for fn, ofn in ((dyn._forward_lf, orc._forward_lf), (dyn._backward_lf, orc._backward_lf)):
            x1, v1, ld = fn(x, v, beta, step)
            ox, ov, old = ofn(x, v, beta, step)
            assert H.relerr(np_(x1), ox) < TOL_OP and H.relerr(np_(v1), ov) < TOL_OP, (step, fn.__name__)
            assert H.relerr(np_(ld), old) < TOL_OP, (step, fn.__name__)
            assert np.abs(old).max() > 1e-3                      # the log-det path is exercised
    # inverse pair at full width (:537-590)
    x1, v1, ld = dyn._forward_lf(x, v, beta, 3)
    x2, v2, ld2 = dyn._backward_lf(x1, v1, beta, N - 1 - 3)
    assert H.relerr(np_(x2), x) < 2e-5 and H.relerr(np_(v2), v) < 2e-5 and H.relerr(np_(ld2), -np_(ld)) < 2e-5


def test_cfg5_transition_kernel_and_apply_transition_at_full_width(la, cfg5):
    """A 3-step `transition_kernel` in both directions and one `apply_transition` (:195-313, :592-609)
    at D=2048 / H=8192 against the fp64 oracle at the single-op bar."""
    T, X, xp, vp = cfg5
    N, eps, beta, B, D = 3, 0.1, 4.0, 5, 2 * T * X
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D, seed=105)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)     # noqa: E731
    x, v0f, v0b, u = f32(x), f32(v0f), f32(v0b), f32(u)
    for fwd, v0 in ((True, v0f), (False, v0b)):
        xo, vo, p, sld = dyn.transition_kernel(x, beta, forward=fwd, momentum=v0, return_logdet=True)
        want = orc.transition_kernel(x, beta, v0, forward=fwd)
        assert H.relerr(np_(xo), want[0]) < TOL_OP and H.relerr(np_(vo), want[1]) < TOL_OP, fwd
        assert H.relerr(np_(sld), want[3]) < TOL_OP and np.abs(np_(p) - want[2]).max() < TOL_P, fwd
    want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    for both in (True, False):
        dyn.both_directions = both
        got = [np_(g) for g in dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)]
        assert H.relerr(got[0], want[0]) < TOL_OP and H.relerr(got[1], want[1]) < TOL_OP
        assert np.abs(got[2] - want[2]).max() < TOL_P
        safe = np.abs(want[2] - u) > 1e-4
        assert H.relerr(got[3][safe], want[3][safe]) < TOL_OP


def test_cfg4_conv3d_at_its_real_trajectory_length(la):
    """BASELINE.json configs[3]: 16x16, ConvNet3D F=16 / H=1024, the real N_LF = 15 (eps 0.2, beta 3, hot start),
    a handful of chains: single steps at the first and last index at the 1e-5 bar, the whole 15-step transition
    within the measured fp32 envelope (profiles/r02_error_ratio_*.txt)."""
    L, N, eps, beta, B = 16, 15, 0.2, 3.0, 5
    D = 2 * L * L
    xp, vp = H.conv_weights(L, L, regime="init")
    orc = H.gauge_oracle(L, L, N, eps, xp, vp, arch='conv3D')
    orc32 = H.gauge_oracle(L, L, N, eps, xp, vp, arch='conv3D', dtype=np.float32)
    dyn = H.gauge_hip(L, L, N, eps, xp, vp, orc.mask, B, arch='conv3D')
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D, seed=104)
    for step in (0, N - 1):
        for fn, ofn in ((dyn._forward_lf, orc._forward_lf), (dyn._backward_lf, orc._backward_lf)):
            x1, v1, ld = fn(x, v0f, beta, step)
            ox, ov, old = ofn(x, v0f, beta, step)
            assert H.relerr(np_(x1), ox) < TOL_OP and H.relerr(np_(v1), ov) < TOL_OP and H.relerr(np_(ld), old) < TOL_OP
    want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    f32 = orc32.apply_transition(x.astype(np.float32), beta, v0f.astype(np.float32), v0b.astype(np.float32), coin,
                                 u.astype(np.float32))
    got = [np_(g) for g in dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)]
    assert_fp32_equivalent(got[0], want[0], f32[0], "x_prop")
    assert_fp32_equivalent(got[1], want[1], f32[1], "v_prop")
    assert np.abs(got[2] - want[2]).max() < max(TOL_P, P_RATIO * np.abs(f32[2] - want[2]).max())


def _full_size_properties(dyn, x, v, beta, rev_tol):
    """Size-independent properties of a whole trajectory at a BASELINE per-GPU batch (no oracle run at this size):
    (1) the backward trajectory undoes the forward one and the log-dets cancel (gauge_dynamics.py:537-590),
    (2) the launch is deterministic, (3) accept probabilities lie in [0, 1], (4) a whole MCMC step returns, chain by
    chain, either its proposal or its input, wrapped observables are finite."""
    x1, v1, p1, ld1 = dyn.transition_kernel(x, beta, forward=True, momentum=v, return_logdet=True)
    x2, v2, p2, ld2 = dyn.transition_kernel(x1, beta, forward=False, momentum=v1, return_logdet=True)
    # rev_tol = (rms, max) bounds on the round trip: 2N chaotic steps amplify fp32 rounding, the tail of millions of
    # elements more than their bulk.  Measured rms / max relative error of v (the worst of x, v, logdet):
    # cfg 3 (20 steps) 1.8e-5 / 3.6e-4, cfg 4 (30) 7.7e-5 / 1.6e-3, cfg 5 (50) 2.0e-4 / 5.9e-3.
    rms_tol, max_tol = rev_tol
    for got, want in ((x2, x), (v2, v), (ld2, -ld1)):
        d = (got - want).double()
        scale = max(1.0, float(want.double().pow(2).mean().sqrt()))
        assert float(d.pow(2).mean().sqrt()) / scale < rms_tol
        assert H.relerr(np_(got), np_(want)) < max_tol
    x1b, _, p1b, _ = dyn.transition_kernel(x, beta, forward=True, momentum=v, return_logdet=True)
    assert torch.equal(x1, x1b) and torch.equal(p1, p1b)
    assert torch.isfinite(x1).all() and torch.all((p1 >= 0) & (p1 <= 1)) and torch.all((p2 >= 0) & (p2 <= 1))
    out = dyn(x, beta)
    same = (out[3] == x).all(1) | (out[3] == out[0]).all(1)
    assert bool(same.all())
    return x1, v1, p1


def test_full_size_properties_cfg4(la):
    """BASELINE.json configs[3] at its per-GPU size: 16x16, ConvNet3D F=16 / H=1024, 1024 chains (8192 over 8
    GPUs), 15 LF steps, eps 0.2, beta 3.  The first 4 chains are also held against the oracle."""
    L, N, eps, beta, B = 16, 15, 0.2, 3.0, 1024
    D = 2 * L * L
    xp, vp = H.conv_weights(L, L, regime="init")
    orc = H.gauge_oracle(L, L, N, eps, xp, vp, arch='conv3D')
    orc32 = H.gauge_oracle(L, L, N, eps, xp, vp, arch='conv3D', dtype=np.float32)
    dyn = H.gauge_hip(L, L, N, eps, xp, vp, orc.mask, B, arch='conv3D')
    torch.manual_seed(4)
    x = torch.rand(B, D, device="cuda") * (2 * np.pi)
    v = torch.randn(B, D, device="cuda")
    x1, v1, p1 = _full_size_properties(dyn, x, v, beta, (3e-4, 5e-3))
    xs, vs = np_(x[:4]), np_(v[:4])
    want = orc.transition_kernel(xs, beta, vs, forward=True)
    w32 = orc32.transition_kernel(xs.astype(np.float32), beta, vs.astype(np.float32), forward=True)
    assert_fp32_equivalent(np_(x1[:4]), want[0], w32[0], "x (first 4 chains)")
    assert_fp32_equivalent(np_(v1[:4]), want[1], w32[1], "v (first 4 chains)")


def test_full_size_properties_cfg5(la, cfg5):
    """BASELINE.json configs[4] at its per-GPU size: 32x32, GenericNet D=2048 / H=8192, 2048 chains (16384 over 8
    GPUs), 25 LF steps, eps 0.1, beta 4 -- 4096 rows x 50 network calls of 151 M parameters per launch."""
    T, X, xp, vp = cfg5
    N, eps, beta, B, D = 25, 0.1, 4.0, 2048, 2 * T * X
    masks = H.gauge_oracle(T, X, N, eps, xp, vp).mask
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, masks, B)
    torch.manual_seed(5)
    x = torch.rand(B, D, device="cuda") * (2 * np.pi)
    v = torch.randn(B, D, device="cuda")
    _full_size_properties(dyn, x, v, beta, (8e-4, 2e-2))


def test_chain_statistics_on_device_histories(la):
    """f4: the estimators behind ESS/sec and tau_int (utils/func_utils.py:45-54,114-120, utils/autocorr.py:23-199)
    fed a DEVICE tensor history -- what a device-resident run keeps in HBM -- against the lag-sum restatements of
    oracle/stats.py (independent of any FFT)."""
    from oracle import stats as ostats
    from l2hmc_amd import stats
    rng = np.random.default_rng(11)
    n, chains, dims = 160, 6, 5
    e = rng.standard_normal((n, chains, dims))
    x = np.zeros_like(e)
    for t in range(1, n):                                   # AR(1): known autocorrelation 0.8^lag
        x[t] = 0.8 * x[t - 1] + e[t]
    xd = torch.as_tensor(x, device="cuda")
    np.testing.assert_allclose(stats.acl_spectrum(xd, 2.0), ostats.acl_spectrum_direct(x, 2.0), rtol=1e-9, atol=1e-11)
    for tau in (0, 1, 17):
        assert abs(stats.autocovariance(xd, tau) - ostats.autocovariance_direct(x, tau)) < 1e-10
    A = stats.acl_spectrum(xd, np.sqrt(stats.autocovariance(xd, 0)))
    assert abs(stats.ESS(A) - stats.ESS(ostats.acl_spectrum_direct(x, np.sqrt(ostats.autocovariance_direct(x, 0))))) < 1e-9
    np.testing.assert_allclose(stats.autocorr(xd[:, 0, 0]), ostats.autocorr_direct(x[:, 0, 0]), atol=1e-10)
    np.testing.assert_allclose(stats.autocorr_func_1d(xd[:, 1, 2]), ostats.acf_direct(x[:, 1, 2]), atol=1e-10)
    np.testing.assert_allclose(stats.autocorr_fast(xd[:, 2, 1], kappa=60), ostats.acf_direct(x[:, 2, 1], unbiased=True)[:60],
                               atol=1e-10)
    tau_d, _ = stats.integrated_time(xd[:, :, :2], quiet=True)
    for d in range(2):
        assert tau_d[d] == pytest.approx(ostats.integrated_time_direct(x[:, :, d]), rel=1e-9)
    # and on a real sampler history: float32 samples straight from the device-resident loop
    T = X = 4
    xp, vp = H.gauge_weights(T, X, regime="init")
    orc = H.gauge_oracle(T, X, 3, 0.1, xp, vp)
    dyn = H.gauge_hip(T, X, 3, 0.1, xp, vp, orc.mask, 16)
    smp = la.GaugeSampler(dyn)
    xs, hist = torch.rand(16, 32, device="cuda") * 6.28, []
    for _ in range(24):
        xs = smp.step(xs, 2.0)[0]
        hist.append(xs)
    Hd = torch.stack(hist)                                   # [steps, chains, dims] on the device
    feats = torch.cat([torch.cos(Hd), torch.sin(Hd)], dim=2)
    got = stats.acl_spectrum(feats, 1.0)
    want = ostats.acl_spectrum_direct(feats.cpu().numpy().astype(np.float64), 1.0)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("L,arch,B,both", [(8, "generic", 37, True), (8, "generic", 37, False), (8, "conv3D", 12, True),
                                           (6, "generic", 9, True)])
def test_dynamics_call_with_library_draws_matches_oracle(la, L, arch, B, both):
    """`dynamics(x, beta)` as the reference calls it (gauge_model.py:753,844): no draw injected, so the library
    draws for itself (l2hmc_gauge_transition_draw: one launch on plans with a whole-trajectory kernel).  The Philox
    streams are reproducible through l2hmc_fill_*, so the oracle can be fed the very same draws."""
    from l2hmc_amd import _lib
    N, eps, beta, D = 3, 0.1, 2.0, 2 * L * L
    xp, vp = (H.conv_weights if arch == "conv3D" else H.gauge_weights)(L, L, regime="mild")
    orc = H.gauge_oracle(L, L, N, eps, xp, vp, arch=arch)
    dyn = H.gauge_hip(L, L, N, eps, xp, vp, orc.mask, B, arch=arch, both_directions=both)
    x0 = np.random.default_rng(3).uniform(0, 2 * np.pi, (B, D)).astype(np.float32)
    dyn._draws = 5                                           # -> stream pair (6, 7)
    got = [np_(g) for g in dyn(x0, beta)]
    assert dyn._draws == 8
    V = torch.empty(2 * B, D, device="cuda")
    cu = torch.empty(2 * B, device="cuda")
    _lib.check(_lib.lib().l2hmc_fill_normal(V.data_ptr(), V.numel(), dyn._seed, 6, None))
    _lib.check(_lib.lib().l2hmc_fill_uniform(cu.data_ptr(), cu.numel(), dyn._seed, 7, None))
    V, cu = np_(V), np_(cu)
    want = orc.apply_transition(x0.astype(np.float64), beta, V[:B], V[B:], cu[:B], cu[B:])
    assert H.relerr(got[0], want[0]) < 2 * TOL_OP and H.relerr(got[1], want[1]) < 2 * TOL_OP
    assert np.abs(got[2] - want[2]).max() < TOL_P
    safe = np.abs(want[2] - cu[B:]) > 1e-4
    assert H.relerr(got[3][safe], want[3][safe]) < 2 * TOL_OP
    acc = got[2] > cu[B:]
    np.testing.assert_array_equal(got[3][acc & safe], got[0][acc & safe])
    np.testing.assert_array_equal(got[3][~acc & safe], x0[~acc & safe])


# ----------------------------------------------------------------- kept first-layer products (layer-by-layer path)
@pytest.mark.parametrize("L,arch,N,B", [(8, "generic", 4, 37), (4, "generic", 3, 70), (8, "conv3D", 3, 21),
                                        (16, "conv3D", 3, 9), (16, "generic", 2, 130), (6, "generic", 3, 11)])
def test_kept_products_equal_recomputed(la, L, arch, N, B):
    """The layer-by-layer path keeps the first-layer products a leapfrog step repeats (XNet's product with the
    momentum across the two position sub-updates, VNet's whole product -- and the force -- from the end of one step
    to the start of the next; csrc/leapfrog.hip).  An fp32 fma chain cut at a tile boundary and continued later has
    the bits of the uninterrupted chain: every output must EQUAL the recomputing path (L2HMC_PLAN_RECOMPUTE, the
    reference's evaluation order, gauge_dynamics.py:412-483).  6x6 (x_dim 72: ragged tiles) has nothing to keep and
    must simply agree with itself."""
    mk = H.conv_weights if arch == "conv3D" else H.gauge_weights
    xp, vp = mk(L, L, regime="mild")
    orc = H.gauge_oracle(L, L, N, 0.15, xp, vp, arch=arch)
    dyn = H.gauge_hip(L, L, N, 0.15, xp, vp, orc.mask, B, arch=arch)
    dyn.fused = False
    x, v0f, v0b, coin, u = H.gauge_inputs(B, 2 * L * L, seed=311)
    outs = {}
    for rec in (False, True):
        dyn.recompute = rec
        o = [dyn.transition_kernel(x, 2.0, forward=fwd, momentum=v0, return_logdet=True)
             for fwd, v0 in ((True, v0f), (False, v0b))]
        o.append(dyn.apply_transition(x, 2.0, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u))
        outs[rec] = [t for tup in o for t in tup]
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)
    # and the kept-product path is a correct trajectory (forward, against the oracle)
    want = orc.transition_kernel(x, 2.0, v0f, forward=True)
    assert H.relerr(np_(outs[False][0]), want[0]) < 2 * TOL_OP and H.relerr(np_(outs[False][1]), want[1]) < 2 * TOL_OP


# ----------------------------------------------------------------- heads on the active columns only (layer-by-layer path)
@pytest.mark.parametrize("L,arch,N,B", [(8, "generic", 4, 128), (8, "generic", 3, 192), (16, "conv3D", 3, 64),
                                        (16, "generic", 2, 256), (8, "conv3D", 3, 64), (8, "generic", 3, 37)])
def test_active_column_heads_equal_all_columns(la, L, arch, N, B):
    """A position sub-update moves only the columns its keep mask does not hold fixed (gauge_dynamics.py:519-531,
    :574-584: x' = keep x + (1 - keep) (...), the log-det term carries the same factor), so where the rows' directions
    are known per row tile (apply_transition: rows [0, B) forward, [B, 2B) backward; forward-only trajectories) the
    layer-by-layer path forms S / T / Q for those columns alone (csrc/stq_dense.hip: HeadsArgs::cols_f, lists from
    active_cols_kernel).  A column's dot product keeps its k order, so x, v and the accept probability's inputs must
    EQUAL the all-columns evaluation; the log-det is the same terms added tile by tile in another grouping (fp32
    rounding).  B = 37 (rows not whole tiles: the split is not on a tile edge) takes the all-columns path either way."""
    from l2hmc_amd import _lib
    mk = H.conv_weights if arch == "conv3D" else H.gauge_weights
    xp, vp = mk(L, L, regime="mild")
    orc = H.gauge_oracle(L, L, N, 0.15, xp, vp, arch=arch)
    dyn = H.gauge_hip(L, L, N, 0.15, xp, vp, orc.mask, B, arch=arch)
    dyn.fused = False
    x, v0f, v0b, coin, u = H.gauge_inputs(B, 2 * L * L, seed=313)
    outs = {}
    try:
        for on in (1, 0):
            dyn.all_columns = not on                     # L2HMC_PLAN_ALL_COLUMNS
            f = dyn.transition_kernel(x, 2.0, forward=True, momentum=v0f, return_logdet=True)
            tr = dyn.apply_transition(x, 2.0, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
            lf = dyn._forward_lf(x, v0f, 2.0, 1)
            outs[on] = (f, tr, lf)
    finally:
        dyn.all_columns = False
    (f1, tr1, lf1), (f0, tr0, lf0) = outs[1], outs[0]
    assert torch.equal(f1[0], f0[0]) and torch.equal(f1[1], f0[1])                  # x, v of the forward trajectory
    assert torch.equal(lf1[0], lf0[0]) and torch.equal(lf1[1], lf0[1])              # one leapfrog step
    assert torch.equal(tr1[0], tr0[0]) and torch.equal(tr1[1], tr0[1]) and torch.equal(tr1[3], tr0[3])
    scale = max(1.0, float(f0[3].abs().max()))
    assert float((f1[3] - f0[3]).abs().max()) <= 2e-6 * scale                      # sum of log-dets: grouping only
    assert float((lf1[2] - lf0[2]).abs().max()) <= 2e-6 * max(1.0, float(lf0[2].abs().max()))
    assert float((f1[2] - f0[2]).abs().max()) <= 1e-5 and float((tr1[2] - tr0[2]).abs().max()) <= 1e-5   # p_accept
    want = orc.transition_kernel(x, 2.0, v0f, forward=True)
    assert H.relerr(np_(f1[0]), want[0]) < 2 * TOL_OP and H.relerr(np_(f1[1]), want[1]) < 2 * TOL_OP


def test_active_column_heads_in_the_128_row_tile_form_with_a_direction_split(la):
    """`heads32_kernel` (128-row x 64-column tiles, grids of >= 512 tiles: the cfg-5 path) in its
    active-column form WITH apply_transition's [forward B | backward B] row layout, where tiles at or beyond
    `dir_split` take the backward list (cols_b / cnt_b).  8x8 lattice, B = 16384 chains -> 2 B = 32768 rows = 512 tiles
    of 128 x 64 with the split on a tile edge.  Against the all-columns evaluation (L2HMC_PLAN_ALL_COLUMNS): x, v, x_out
    EQUAL, p within 1e-5; the first 64 chains against the float64 oracle.  A second batch (B = 16448: split % 128 == 64,
    where the library itself must fall back to all columns) goes through the same checks."""
    T = X = 8
    N, eps, beta = 2, 0.15, 2.0
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    outs = {}
    for B in (16384, 16448):
        dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
        dyn.fused = False
        x, v0f, v0b, coin, u = H.gauge_inputs(B, 2 * T * X, seed=331)
        for allc in (False, True):
            dyn.all_columns = allc
            outs[B, allc] = dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
        a, b = outs[B, False], outs[B, True]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]), B
        assert float((a[2] - b[2]).abs().max()) <= 1e-5, B
        want = orc.apply_transition(x[:64], beta, v0f[:64], v0b[:64], coin[:64], u[:64])
        assert H.relerr(np_(a[0][:64]), want[0]) < 2 * TOL_OP and H.relerr(np_(a[1][:64]), want[1]) < 2 * TOL_OP
        assert np.abs(np_(a[2][:64]) - want[2]).max() < TOL_P
        assert float(a[2].mean()) > 0.01                      # not a trivially rejected batch


def test_active_column_heads_with_any_mask(la):
    """The column lists are built from the masks as they are (csrc/stq_dense.hip: active_cols_kernel keeps a column
    wherever keep != 1): rows with 30 % / 80 % ones, an all-ones and an all-zeros row (one of the two sub-updates then
    has NO active column: every heads workgroup of that launch leaves at once) and fractional entries (the column stays
    in both lists).  Against the float64 oracle with the same masks, and equal to the all-columns evaluation."""
    from l2hmc_amd import _lib
    from oracle.gauge_dynamics import GaugeDynamicsOracle
    T = X = 8
    D, N, B = 128, 5, 128
    rng = np.random.default_rng(5)
    masks = np.zeros((N, D))
    masks[0, rng.permutation(D)[:38]] = 1.0
    masks[1, rng.permutation(D)[:102]] = 1.0
    masks[2, :] = 1.0
    masks[3, :] = 0.0
    masks[4, rng.permutation(D)[:64]] = 1.0
    masks[4, rng.permutation(D)[:9]] = 0.5
    masks[0, 7] = 0.25
    xp, vp = H.gauge_weights(T, X, regime="mild")
    orc = GaugeDynamicsOracle(T, X, N, 0.1, masks, xp, vp, "generic")
    dyn = H.gauge_hip(T, X, N, 0.1, xp, vp, masks, B)
    dyn.fused = False
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D, seed=317)
    outs = {}
    try:
        for on in (1, 0):
            dyn.all_columns = not on
            outs[on] = dyn.apply_transition(x, 2.0, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
    finally:
        dyn.all_columns = False
    for a, b in ((outs[1][0], outs[0][0]), (outs[1][1], outs[0][1]), (outs[1][3], outs[0][3])):
        assert torch.equal(a, b)
    assert float((outs[1][2] - outs[0][2]).abs().max()) <= 1e-5
    want = orc.apply_transition(x, 2.0, v0f, v0b, coin, u)
    assert H.relerr(np_(outs[1][0]), want[0]) < 2 * TOL_OP and H.relerr(np_(outs[1][1]), want[1]) < 2 * TOL_OP
    assert np.abs(np_(outs[1][2]) - want[2]).max() < TOL_P
    assert H.relerr(np_(outs[1][3]), want[3]) < 2 * TOL_OP


# ----------------------------------------------------------------- sub-tile form of the whole-trajectory kernel
@pytest.mark.parametrize("B", [3, 130, 256, 500, 700, 1024, 1100, 1536, 2049, 2304, 3000, 3600, 4096, 4100, 5000, 6000, 6200])
def test_subtile_and_32_row_forms_equal_16_row_form(la, B):
    """Batches of more than one round of 16-row workgroups run in the 32-row form (csrc/fused_traj32.hip: every weight
    fragment feeds two MFMAs), cut into up to three launches with a 16-row round and / or a sub-tile launch for the rest
    (launch_fused_step: 3600 and 4096 chains one 32-row launch, 2049 / 3000: 16-row + sub-tile, 4100 / 5000: 32-row +
    sub-tile, 6000: 32-row + 16-row, 6200: all three).  Batches that cannot put a 16-row tile on every CU run the whole-trajectory kernel in its sub-tile form (4, 8 or 12
    rows per workgroup on v_mfma_f32_4x4x1_16B_f32, csrc/fused_traj4.hip).  Same k order, same epilogue expressions,
    same grouping of every sum, same Philox indexing: a whole MCMC step, the trajectories alone (both directions, per-row
    directions) and single leapfrog steps must EQUAL the 16-row form bit for bit; and the sub-tile form agrees with
    the oracle on its own."""
    from l2hmc_amd import _lib, GaugeSampler
    T = X = 8
    N, eps, beta = 4, 0.2, 2.0
    orc, orc32, dyn = _pair(T, X, N, eps, B, "mild", True)
    rng = np.random.default_rng(21)
    x = torch.as_tensor(rng.uniform(0, 2 * np.pi, (B, 128)), dtype=torch.float32, device="cuda")
    v = torch.as_tensor(rng.standard_normal((B, 128)), dtype=torch.float32, device="cuda")
    coin = torch.as_tensor(rng.uniform(size=B), dtype=torch.float32, device="cuda")
    uu = torch.as_tensor(rng.uniform(size=B), dtype=torch.float32, device="cuda")
    outs = {}
    try:
        for sub in (1, 0):
            dyn.tiles16_only = not sub                   # L2HMC_PLAN_TILES16_ONLY
            dyn._draws = 40
            smp = GaugeSampler(dyn)
            xn, px, obs, dq = smp.step(x, beta)
            dyn._draws = 40
            tr = dyn.apply_transition(x, beta)                                   # library draws: the step kernel
            f = dyn.transition_kernel(x, beta, forward=True, momentum=v, return_logdet=True)
            b = dyn.transition_kernel(x, beta, forward=False, momentum=v, return_logdet=True)
            lf = dyn._forward_lf(x, v, beta, 2) + dyn._backward_lf(x, v, beta, 1)
            # injected draws: both directions of the batch in one trajectory launch (rows = 2 B, x_mod / dir_split)
            inj = dyn.apply_transition(x, beta, momentum_f=v, momentum_b=v.flip(0), coin=coin, u=uu)
            dyn.both_directions = False
            dyn._draws = 40
            sel = GaugeSampler(dyn).step(x, beta)[:2]
            dyn.both_directions = True
            outs[sub] = [xn, px, obs["action"], obs["avg_plaq"], obs["top_charge"], dq, *tr, *f, *b, *lf, *sel, *inj,
                         smp.stats.mean_accept()]
    finally:
        dyn.tiles16_only = False
    for i, (a, b_) in enumerate(zip(outs[1][:-1], outs[0][:-1])):
        assert torch.equal(a, b_), f"output {i} differs between the sub-tile and the 16-row form"
    # the step's mean accept probability is a fixed-order sum of per-WORKGROUP partial sums: another grouping of the
    # same per-chain values (which are equal), so it agrees to fp32 rounding
    assert abs(outs[1][-1] - outs[0][-1]) <= 1e-6 * max(abs(outs[0][-1]), 1e-30) + 1e-12
    want = orc.transition_kernel(np_(x), beta, np_(v), forward=True)
    f32 = orc32.transition_kernel(np_(x).astype(np.float32), beta, np_(v).astype(np.float32), forward=True)
    got = outs[1][10:14]
    assert H.relerr(np_(got[0]), want[0]) < 2 * TOL_OP and H.relerr(np_(got[1]), want[1]) < 2 * TOL_OP
    assert np.abs(np_(got[2]) - want[2]).max() < max(TOL_P, P_RATIO * np.abs(f32[2] - want[2]).max())
