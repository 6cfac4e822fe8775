"""Regenerates the committed training fixtures (tests/golden/train_*.npz): loss and gradients of the training
graph on fixed seeded inputs, produced by the repo's own float64 differentiable oracle (oracle/torch_ref.py;
torch.autograd stands in for tf.gradients).  As with make_golden.py these are pins of the oracle and fixed
inputs / expected outputs for the HIP parity tests -- the reference cannot run here and ships no such vectors.
Weights come from the seeded initialisers of tests/helpers.py (checksums stored); large gradient matrices are
stored as [sum, sum |.|, sum of squares] plus their first row, small ones in full.

    python tests/golden/make_golden_train.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import dynamics as od  # noqa: E402
from oracle.torch_ref import TorchGaugeModel, TorchDynamicsModel  # noqa: E402
from tests import helpers as H  # noqa: E402

BIG = 4096          # arrays with more entries are summarised


def summarise(prefix, grads, out):
    for k, g in sorted(grads.items()):
        g = np.asarray(g, dtype=np.float64)
        if g.size > BIG:
            out[f"{prefix}/{k}/stats"] = np.array([g.sum(), np.abs(g).sum(), (g * g).sum()])
            out[f"{prefix}/{k}/row0"] = g.reshape(g.shape[0], -1)[0]
        else:
            out[f"{prefix}/{k}"] = g


def inputs(B, D, seed, gauge=True):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 2 * np.pi, (B, D)) if gauge else None
    z = rng.standard_normal((B, D))
    mk = lambda: (rng.standard_normal((B, D)), rng.standard_normal((B, D)),   # noqa: E731
                  rng.uniform(size=B) if gauge else rng.integers(0, 2, B).astype(np.float64), rng.uniform(size=B))
    return rng, x, z, mk(), mk()


def gauge_case(name, arch, N, eps, beta, B, regime):
    T = X = 8
    xp, vp = (H.gauge_weights if arch == 'generic' else H.conv_weights)(T, X, seed=106, regime=regime)
    orc = H.gauge_oracle(T, X, N, eps, xp, vp, arch=arch)
    tm = TorchGaugeModel(T, X, N, eps, orc.mask, xp, vp, arch=arch)
    _, x, z, dx, dz = inputs(B, 2 * T * X, 7)
    tt = lambda a: torch.tensor(a, dtype=torch.float64)   # noqa: E731
    loss, terms = tm.loss(tt(x), tt(z), beta, tuple(map(tt, dx)), tuple(map(tt, dz)))
    loss.backward()
    out = dict(T=T, X=X, num_steps=N, eps=eps, beta=beta, arch=arch, regime=regime, masks=orc.mask, x=x, z=z,
               loss=float(loss.detach()), terms=terms.detach().numpy(), grad_eps=float(tm.eps.grad))
    for i, a in enumerate(dx):
        out[f"draws_x/{i}"] = a
    for i, a in enumerate(dz):
        out[f"draws_z/{i}"] = a
    summarise("xnet", {k: v.grad.numpy() for k, v in tm.xnet.items()}, out)
    summarise("vnet", {k: v.grad.numpy() for k, v in tm.vnet.items()}, out)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", out["loss"], "d/deps", out["grad_eps"])


def mog_case(name, N, eps, B, regime):
    tgt = H.mog_target_oracle()
    xp, vp = H.mlp_weights(2, 50, seed=106, regime=regime)
    masks = od.make_masks(N, 2, np.random.RandomState(3))
    tm = TorchDynamicsModel(tgt, N, eps, masks, xp, vp)
    rng, _, z, dx, dz = inputs(B, 2, 11, gauge=False)
    x = tgt.get_samples(B, rng)
    tt = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))   # noqa: E731
    loss, Lx, px, Lz, pz = tm.mog_loss(tt(x), tt(z), tuple(map(tt, dx)), tuple(map(tt, dz)), 0.1)
    loss.backward()
    out = dict(trajectory_length=N, eps=eps, num_nodes=50, scale=0.1, regime=regime, masks=masks, x=x, z=z,
               loss=float(loss.detach()), Lx=Lx.detach().numpy(), px=px.detach().numpy(), Lz=Lz.detach().numpy(),
               pz=pz.detach().numpy(), grad_alpha=float(tm.alpha.grad))
    for i, a in enumerate(dx):
        out[f"draws_x/{i}"] = a
    for i, a in enumerate(dz):
        out[f"draws_z/{i}"] = a
    summarise("xnet", {k: v.grad.numpy() for k, v in tm.xnet.items()}, out)
    summarise("vnet", {k: v.grad.numpy() for k, v in tm.vnet.items()}, out)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", out["loss"], "d/dalpha", out["grad_alpha"])


if __name__ == "__main__":
    gauge_case("train_L8_generic", 'generic', 3, 0.1, 2.5, 6, "mild")
    gauge_case("train_L8_conv3d", 'conv3D', 2, 0.1, 2.5, 6, "mild")
    mog_case("train_mog", 5, 0.1, 12, "stress")
