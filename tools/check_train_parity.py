"""One-off evidence (not a test): gradient parity of the training step at the benchmark trajectory length
(8x8, GenericNet H=512, 10 LF) against float64 torch.autograd of the oracle graph, on a 256-chain sample.
Uses tests/ infrastructure (oracle as the checker).   python tools/check_train_parity.py [B] [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_train import _setup, _ref_grads, _packed_ref  # noqa: E402
from tests import helpers as H  # noqa: E402
from oracle.torch_ref import TorchGaugeModel  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # the GPU box grants a 16-CPU share
    for regime, eps in (("init", 0.25), ("mild", 0.1)):
        print(f"[{regime}] building ...", flush=True)
        tr, tm, x, z, dx, dz = _setup(8, N, eps, B, regime)
        t0 = time.perf_counter()
        loss, *_ = tr.calc_loss_and_grads(x, 2.0, z=z, draws_x=dx, draws_z=dz)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        want, _ = _ref_grads(tm, x, z, dx, dz, 2.0, 'cos_diff')
        t2 = time.perf_counter()
        gv = tr.grad_views()
        worst = {}
        for name, net in (("xnet", tm.xnet), ("vnet", tm.vnet)):
            for k, w in _packed_ref(net).items():
                got = gv[name][k].cpu().numpy().astype(np.float64).reshape(w.shape)
                worst[f"{name}.{k}"] = float(np.abs(got - w).max() / np.abs(w).max())
        worst["eps"] = abs(float(gv["eps"][0]) - float(tm.eps.grad)) / abs(float(tm.eps.grad))
        # the same graph in float32 torch (the reference's precision): its distance from float64 is the yardstick
        xp, vp = H.gauge_weights(8, 8, regime=regime)
        t32 = TorchGaugeModel(8, 8, N, eps, tm.mask.numpy(), xp, vp, dtype=torch.float32)
        f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)   # noqa: E731
        l32, _ = t32.loss(f32(x), f32(z), 2.0, tuple(map(f32, dx)), tuple(map(f32, dz)))
        l32.backward()
        yard = {}
        for name, net, n32 in (("xnet", tm.xnet, t32.xnet), ("vnet", tm.vnet, t32.vnet)):
            ref64, ref32 = _packed_ref(net), _packed_ref(n32)
            for k in ref64:
                yard[f"{name}.{k}"] = float(np.abs(ref32[k] - ref64[k]).max() / np.abs(ref64[k]).max())
        yard["eps"] = abs(float(t32.eps.grad) - float(tm.eps.grad)) / abs(float(tm.eps.grad))
        print(f"[{regime}, eps {eps}, B {B}, {N} LF] loss HIP {float(loss):.6f} vs fp64 {want:.6f}; "
              f"HIP {1e3 * (t1 - t0):.1f} ms (first call), fp64 autograd {t2 - t1:.1f} s on {torch.get_num_threads()} threads")
        print("   max |dg| / max |g| per tensor:", {k: f"{v:.1e}" for k, v in worst.items()})
        print(f"   worst {max(worst.values()):.2e}", flush=True)
        print("   float32 torch autograd of the same graph vs fp64:", {k: f"{v:.1e}" for k, v in yard.items()})
        print(f"   worst {max(yard.values()):.2e}   (HIP / fp32-torch ratio of the worst tensor: "
              f"{max(worst.values()) / max(yard.values()):.2f})", flush=True)


if __name__ == "__main__":
    main()
