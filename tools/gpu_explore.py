"""Developer script (not a test): prints HIP-vs-oracle errors and rough timings
on a GPU box.  `python tests/gpu_explore.py > gpurun_out/explore.txt`."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as H  # noqa: E402
from oracle import lattice as olat, nets as onets, dynamics as ogen  # noqa: E402
import l2hmc_amd as la  # noqa: E402


def np_(t):
    return t.detach().cpu().numpy().astype(np.float64)


def main():
    print("device", torch.cuda.get_device_name(0), flush=True)
    # ---- u1
    for (T, X, B) in [(8, 8, 7), (16, 16, 5), (32, 32, 3), (4, 6, 9), (17, 17, 2)]:
        x = np.random.default_rng(1).uniform(-7, 7, (B, 2 * T * X)).astype(np.float32)
        o = la.u1_observables(x, T, X, beta=2.5, want_force=True)
        x64 = x.astype(np.float64)
        print(f"u1 {T}x{X} B={B}: action {H.relerr(np_(o['action']), olat.total_action(x64, T, X)):.2e} "
              f"force {H.relerr(np_(o['force']), 2.5 * olat.grad_action(x64, T, X)):.2e} "
              f"plaq {H.relerr(np_(o['avg_plaq']), olat.avg_plaq(x64, T, X)):.2e} "
              f"Q {H.relerr(np_(o['top_charge']), olat.top_charge(x64, T, X)):.2e}", flush=True)

    # ---- stq dense
    T = X = 8
    D = 128
    for regime in ("init", "stress"):
        xp, vp = H.gauge_weights(T, X, regime=regime)
        net = la.GenericNet(model_name='XNet', x_dim=D, num_hidden=4 * D, factor=2., name_scope='position',
                            links_shape=(T, X, 2))
        net.load_state(xp)
        rng = np.random.default_rng(5)
        a, b = rng.standard_normal((100, D)), rng.uniform(0, 6.3, (100, D))
        t = np.array([[np.cos(0.7), np.sin(0.7)]])
        S, Tt, Q = net([a, b, t])
        So, To, Qo = onets.generic_net(xp, [a, b, np.tile(t, (100, 1))])
        print(f"stq {regime}: S {H.relerr(np_(S), So):.2e} (max {np.abs(So).max():.2e}) T {H.relerr(np_(Tt), To):.2e} "
              f"Q {H.relerr(np_(Q), Qo):.2e}; abs S {np.abs(np_(S)-So).max():.2e} T {np.abs(np_(Tt)-To).max():.2e}",
              flush=True)

    # ---- leapfrog / trajectory / transition, cfg-3 shape
    for regime, fused in (("init", True), ("init", False), ("mild", True), ("mild", False), ("stress", True)):
        for (T, X, N, eps, beta, B) in [(8, 8, 10, 0.25, 2.0, 71), (4, 4, 3, 0.2, 2.5, 10)]:
            if regime == "stress" and T == 8:
                N = 3
            D = 2 * T * X
            xp, vp = H.gauge_weights(T, X, regime=regime)
            orc = H.gauge_oracle(T, X, N, eps, xp, vp)
            orc32 = H.gauge_oracle(T, X, N, eps, xp, vp, dtype=np.float32)
            dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B)
            dyn.fused = fused
            regime_ = regime
            regime = f"{regime_} fused={fused}"
            x, v0f, v0b, coin, u = H.gauge_inputs(B, D)
            x1, v1, ld = dyn._forward_lf(x, v0f, beta, 1)
            ox, ov, old = orc._forward_lf(x, v0f, beta, 1)
            print(f"[{regime} {T}x{X}] fwd_lf: x {H.relerr(np_(x1), ox):.2e} v {H.relerr(np_(v1), ov):.2e} "
                  f"ld {H.relerr(np_(ld), old):.2e} (|ld| {np.abs(old).max():.2e})", flush=True)
            x1, v1, ld = dyn._backward_lf(x, v0f, beta, 1)
            ox, ov, old = orc._backward_lf(x, v0f, beta, 1)
            print(f"[{regime} {T}x{X}] bwd_lf: x {H.relerr(np_(x1), ox):.2e} v {H.relerr(np_(v1), ov):.2e} "
                  f"ld {H.relerr(np_(ld), old):.2e}", flush=True)
            for fwd in (True, False):
                xo, vo, p, sld = dyn.transition_kernel(x, beta, forward=fwd, momentum=v0f, return_logdet=True)
                a = orc.transition_kernel(x, beta, v0f, forward=fwd)
                a32 = orc32.transition_kernel(x.astype(np.float32), beta, v0f.astype(np.float32), forward=fwd)
                print(f"[{regime} {T}x{X}] traj fwd={fwd}: x {H.relerr(np_(xo), a[0]):.2e} v {H.relerr(np_(vo), a[1]):.2e} "
                      f"p {np.abs(np_(p) - a[2]).max():.2e} sld {H.relerr(np_(sld), a[3]):.2e} | oracle32-vs-64: "
                      f"x {H.relerr(a32[0], a[0]):.2e} p {np.abs(a32[2] - a[2]).max():.2e}  mean p {a[2].mean():.3f}",
                      flush=True)
            want = orc.apply_transition(x, beta, v0f, v0b, coin, u)
            for both in (True, False):
                dyn.both_directions = both
                got = dyn.apply_transition(x, beta, momentum_f=v0f, momentum_b=v0b, coin=coin, u=u)
                print(f"[{regime} {T}x{X}] transition both={both}: " +
                      " ".join(f"{n} {H.relerr(np_(g), w):.2e}" for n, g, w in zip(("xp", "vp", "p", "xo"), got, want)),
                      flush=True)
            regime = regime_

    # ---- MoG / SCG
    for name, tgt_o, nh, N in (("mog", H.mog_target_oracle(), 50, 10), ("scg", H.scg_target_oracle(), 10, 5)):
        xp, vp = H.mlp_weights(2, nh)
        masks = ogen.make_masks(N, 2, np.random.RandomState(42))
        orc = ogen.DynamicsOracle(2, tgt_o, N, 0.1, masks, xp, vp)
        if name == "mog":
            tgt = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
        else:
            tgt = la.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
        dyn = la.Dynamics(2, tgt.get_energy_function(), trajectory_length=N, eps=0.1, net_factory=la.network
                          if nh == 50 else (lambda d, scope, factor: la.network(d, scope, factor, num_nodes=nh)))
        dyn.set_masks(masks)
        dyn.XNet.load_state(xp)
        dyn.VNet.load_state(vp)
        rng = np.random.default_rng(102)
        x = tgt_o.get_samples(256, rng)
        v = rng.standard_normal((256, 2))
        e, g = dyn._target.energy_grad(x)
        print(f"{name} energy {H.relerr(np_(e), tgt_o.energy(x)):.2e} grad {H.relerr(np_(g), tgt_o.grad_energy(x)):.2e}")
        for fwd in (True, False):
            fn, ofn = (dyn.forward, orc.forward) if fwd else (dyn.backward, orc.backward)
            X1, V1, p = fn(x, init_v=v)
            a = ofn(x, v)
            print(f"{name} traj fwd={fwd}: x {H.relerr(np_(X1), a[0]):.2e} v {H.relerr(np_(V1), a[1]):.2e} "
                  f"p {np.abs(np_(p) - a[2]).max():.2e} mean p {a[2].mean():.3f}", flush=True)

    # ---- timing, cfg 3
    T = X = 8
    N, eps, beta, B = 10, 0.25, 2.0, 2048
    xp, vp = H.gauge_weights(T, X, regime="init")
    orc = H.gauge_oracle(T, X, N, eps, xp, vp)
    for both, fused in ((True, True), (False, True), (True, False)):
        dyn = H.gauge_hip(T, X, N, eps, xp, vp, orc.mask, B, both_directions=both)
        dyn.fused = fused
        x = torch.rand(B, 128, device="cuda") * 6.28
        for _ in range(3):
            out = dyn(x, beta)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            out = dyn(x, beta)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K
        flops = (2 if both else 1) * B * N * 4726784
        print(f"cfg3 both={both} fused={fused}: {dt*1e3:.3f} ms/transition, useful {B*N/dt/1e6:.2f} M chain-LF/s, "
              f"{flops/dt/1e12:.1f} TFLOP/s, mean p {out[2].mean().item():.3f}", flush=True)


if __name__ == "__main__":
    main()
