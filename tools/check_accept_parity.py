"""One-off evidence (not a test): accept-rate parity over >= 1000 MCMC steps (north_star: +-1 %).  At every step
the float64 oracle sees the device chain's state and the same draws, as in
tests/test_gpu_parity.py::test_accept_rate_parity_on_identical_inputs, but for 1000 steps and two regimes.
    python tools/check_accept_parity.py [steps] [chains] [layered]
`layered`: the layer-by-layer kernels instead of the whole-trajectory kernel (with 64 chains or a multiple, the position
sub-updates then run on the columns they move: csrc/stq_dense.hip, HeadsArgs::cols_f)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402


def run(tag, N, eps, regime, steps, B, layered=False):
    T = X = 8
    xp, vp = H.gauge_weights(T, X, regime=regime)
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    orc32 = H.gauge_oracle(T, X, N, eps, xp, vp, dtype=np.float32)     # the reference's own precision, same inputs
    dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
    dyn.fused = not layered
    rng = np.random.default_rng(11)
    x = rng.uniform(0, 2 * np.pi, (B, 128)).astype(np.float32)
    ph, po, ah, ao, pf = [], [], [], [], []
    t0 = time.time()
    for it in range(steps):
        v0f, v0b = rng.standard_normal((B, 128)), rng.standard_normal((B, 128))
        coin, u = rng.uniform(size=B), rng.uniform(size=B)
        got = dyn.apply_transition(x, 2.0, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
        want = orc.apply_transition(x.astype(np.float64), 2.0, v0f, v0b, coin, u)
        p = got[2].cpu().numpy()
        ph.append(p); po.append(want[2]); ah.append(p > u); ao.append(want[2] > u)
        pf.append(orc32.apply_transition(x, 2.0, v0f.astype(np.float32), v0b.astype(np.float32), coin,
                                         u.astype(np.float32))[2].astype(np.float64))
        x = np.mod(got[3].cpu().numpy(), 2 * np.pi).astype(np.float32)
        if it % 200 == 0:
            print(f"  [{tag}] step {it} ({time.time() - t0:.0f} s)", flush=True)
    ph, po, pf = np.concatenate(ph), np.concatenate(po), np.concatenate(pf)
    print(f"[{tag}] max |p - p_fp64| over all steps and chains: HIP {np.abs(ph - po).max():.2e}, float32 NumPy oracle "
          f"(the reference's precision and op order) {np.abs(pf - po).max():.2e}; rms HIP "
          f"{np.sqrt(np.mean((ph - po) ** 2)):.2e}, float32 oracle {np.sqrt(np.mean((pf - po) ** 2)):.2e}", flush=True)
    print(f"[{tag}: {N} LF, eps {eps}, '{regime}' weights, {steps} steps x {B} chains] mean accept probability "
          f"HIP {ph.mean():.6f}  oracle {po.mean():.6f}  (diff {abs(ph.mean() - po.mean()):.2e}); accepted fraction "
          f"HIP {np.mean(ah):.6f}  oracle {np.mean(ao):.6f}; max |p_hip - p_oracle| {np.abs(ph - po).max():.2e}; "
          f"accept decisions that differ: {int(np.sum(np.concatenate(ah) != np.concatenate(ao)))} of {ph.size}",
          flush=True)


def run_mog(steps, B):
    """BASELINE.json configs[1]: mixture of Gaussians through `propose` (sampler.py:28-59), accept iff p - u >= 0."""
    import l2hmc_amd as la
    from oracle import dynamics as od
    N, eps, nh = 10, 0.1, 50
    tgt_o = H.mog_target_oracle()
    tgt = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
    xp, vp = H.mlp_weights(2, nh, regime="init")
    masks = od.make_masks(N, 2, np.random.RandomState(3))
    orc = od.DynamicsOracle(2, tgt_o, N, eps, masks, xp, vp)
    dyn = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=N, eps=eps,
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=nh))
    dyn.set_masks(masks)
    dyn.XNet.load_state(xp)
    dyn.VNet.load_state(vp)
    rng = np.random.default_rng(12)
    x = tgt_o.get_samples(B, rng).astype(np.float32)
    ph, po, flips = [], [], 0
    for it in range(steps):
        v0f, v0b = rng.standard_normal((B, 2)), rng.standard_normal((B, 2))
        bits, u = rng.integers(0, 2, B).astype(np.float64), rng.uniform(size=B)
        Lx, Lv, px, outs = la.propose(x, dyn, init_v=v0f, do_mh_step=True, init_v_backward=v0b, dir_bits=bits, u=u)
        want = od.propose(x.astype(np.float64), orc, v0f, v0b, bits, u=u, do_mh_step=True)
        p = px.cpu().numpy()
        ph.append(p); po.append(want[2])
        flips += int(np.sum(((p - u) >= 0) != ((want[2] - u) >= 0)))
        x = outs[0].cpu().numpy()
    ph, po = np.concatenate(ph), np.concatenate(po)
    print(f"[MoG cfg 2: 10 LF, eps 0.1, H=50, {steps} steps x {B} chains] mean accept probability HIP {ph.mean():.6f}  "
          f"oracle {po.mean():.6f}  (diff {abs(ph.mean() - po.mean()):.2e}); max |p_hip - p_oracle| "
          f"{np.abs(ph - po).max():.2e}; accept decisions that differ: {flips} of {ph.size}", flush=True)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    layered = len(sys.argv) > 3 and sys.argv[3] == "layered"
    tag = " (layer-by-layer kernels)" if layered else ""
    run("moderate acceptance" + tag, 5, 0.08, "init", steps, B, layered)
    run("benchmark dynamics" + tag, 10, 0.25, "init", steps, B, layered)
    if not layered:
        run_mog(steps, 64)
