"""Regenerates the committed golden fixtures (tests/golden/*.npz).

The reference cannot run in this pipeline (every hot-path module imports
TensorFlow 1.x, which is not installed and cannot be fetched), and
it ships no golden tensors, so these vectors come from the repo's own fp64 CPU
oracle (oracle/), which is pinned against the reference's known answers by
tests/test_oracle_kat.py.  They are regression pins for the oracle and fixed
inputs/outputs for the HIP parity tests -- not reference-produced data.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import lattice as lat, dynamics as gen  # noqa: E402
from tests import helpers as H  # noqa: E402


def wsum(p):
    """Checksums of a weight dict: detect drift of the seeded initialiser."""
    return np.array([[float(np.sum(v)), float(np.sum(np.abs(v)))] for _, v in sorted(p.items())])


def u1_case():
    out = {}
    for (T, X) in ((8, 8), (4, 6)):
        x = np.random.default_rng(11).uniform(-2 * np.pi, 4 * np.pi, (4, 2 * T * X))
        k = f"{T}x{X}"
        out[k + "/x"] = x
        out[k + "/plaq_sums"] = lat.plaq_sums(x, T, X)
        out[k + "/action"] = lat.total_action(x, T, X)
        out[k + "/force_beta2.5"] = 2.5 * lat.grad_action(x, T, X)
        out[k + "/avg_plaq"] = lat.avg_plaq(x, T, X)
        out[k + "/top_charge"] = lat.top_charge(x, T, X)
    np.savez_compressed(os.path.join(HERE, "u1_obs.npz"), **out)


def gauge_case(name, T, X, N, eps, beta, B, regime, store_weights, arch='generic'):
    D = 2 * T * X
    if arch == 'conv3D':
        xp, vp = H.conv_weights(T, X, seed=106, regime=regime)
    else:
        xp, vp = H.gauge_weights(T, X, seed=106, regime=regime)
    orc = H.gauge_oracle(T, X, N, eps, xp, vp, arch=arch)
    x, v0f, v0b, coin, u = H.gauge_inputs(B, D, seed=103)
    out = dict(T=T, X=X, num_steps=N, eps=eps, beta=beta, masks=orc.mask, x=x, v0f=v0f, v0b=v0b, coin=coin, u=u,
               regime=regime, arch=arch, xnet_checksum=wsum(xp), vnet_checksum=wsum(vp))
    if store_weights:
        for k, v in xp.items():
            out["xnet/" + k] = v.astype(np.float32)
        for k, v in vp.items():
            out["vnet/" + k] = v.astype(np.float32)
        xp = {k: v.astype(np.float32).astype(np.float64) for k, v in xp.items()}
        vp = {k: v.astype(np.float32).astype(np.float64) for k, v in vp.items()}
        orc = H.gauge_oracle(T, X, N, eps, xp, vp, arch=arch)
    for tag, v0, fwd in (("f", v0f, True), ("b", v0b, False)):
        trace = []
        xN, vN, p, sld = orc.transition_kernel(x, beta, v0, forward=fwd, trace=trace)
        out[f"traj_{tag}/x_steps"] = np.stack([t[0] for t in trace])
        out[f"traj_{tag}/v_steps"] = np.stack([t[1] for t in trace])
        out[f"traj_{tag}/logdet_steps"] = np.stack([t[2] for t in trace])
        out[f"traj_{tag}/p"] = p
    xpost, vpost, p, xout = orc.apply_transition(x, beta, v0f, v0b, coin, u)
    out.update(x_prop=xpost, v_prop=vpost, p_accept=p, x_out=xout)
    # S/T/Q of the first momentum sub-update of step 0 (position, grad, t) for the kernel-level check
    t0 = orc._format_time(0, tile=B)
    g0 = orc.grad_potential(x, beta)
    S, Tt, Q = orc.momentum_fn([x, g0, t0])
    out.update(stq0_S=S, stq0_T=Tt, stq0_Q=Q, grad0=g0)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def small_case(name, target, num_nodes, N, eps, B, seed):
    xp, vp = H.mlp_weights(2, num_nodes, seed=106, regime="stress")
    xp = {k: v.astype(np.float32).astype(np.float64) for k, v in xp.items()}
    vp = {k: v.astype(np.float32).astype(np.float64) for k, v in vp.items()}
    masks = gen.make_masks(N, 2, np.random.RandomState(42))
    orc = gen.DynamicsOracle(2, target, N, eps, masks, xp, vp)
    rng = np.random.default_rng(seed)
    x = target.get_samples(B, rng)
    v0f, v0b = rng.standard_normal((B, 2)), rng.standard_normal((B, 2))
    bits = rng.integers(0, 2, B).astype(np.float64)
    u = rng.uniform(size=B)
    out = dict(num_nodes=num_nodes, trajectory_length=N, eps=eps, masks=masks, x=x, v0f=v0f, v0b=v0b, dir_bits=bits,
               u=u, energy=target.energy(x), grad_energy=target.grad_energy(x))
    for k, v in xp.items():
        out["xnet/" + k] = v.astype(np.float32)
    for k, v in vp.items():
        out["vnet/" + k] = v.astype(np.float32)
    Xf, Vf, pf = orc.forward(x, v0f)
    Xb, Vb, pb = orc.backward(x, v0b)
    Lx, _, px, outs, Lv = gen.propose(x, orc, v0f, v0b, bits, u, do_mh_step=True)
    out.update(Xf=Xf, Vf=Vf, pf=pf, Xb=Xb, Vb=Vb, pb=pb, Lx=Lx, Lv_mixed=Lv, px=px, x_accept=outs[0])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    u1_case()
    gauge_case("gauge_L4_stress", 4, 4, 3, 0.2, 2.5, 6, "stress", store_weights=True)
    gauge_case("gauge_L8_cfg3_init", 8, 8, 10, 0.25, 2.0, 4, "init", store_weights=False)
    gauge_case("gauge_L8_cfg3_mild", 8, 8, 10, 0.25, 2.0, 4, "mild", store_weights=False)
    gauge_case("gauge_L8_conv3d_mild", 8, 8, 5, 0.25, 2.0, 4, "mild", store_weights=False, arch='conv3D')
    small_case("mog_cfg2", H.mog_target_oracle(), 50, 10, 0.1, 32, 102)
    small_case("scg_cfg1", H.scg_target_oracle(), 10, 5, 0.1, 32, 101)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
