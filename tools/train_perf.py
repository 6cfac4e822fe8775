"""Diagnostic: time of one training step (loss + gradients + Adam) at the benchmark shape.
    python tools/train_perf.py [B] [N]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l2hmc_amd as la  # noqa: E402
from l2hmc_amd.gauge_trainer import GaugeTrainer  # noqa: E402
if os.environ.get("L2HMC_LIB"):          # a diagnostic build (e.g. the previous commit's library, for same-box before/after lines)
    from l2hmc_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["L2HMC_LIB"])


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    arch = sys.argv[4] if len(sys.argv) > 4 else 'generic'
    L = 8
    np.random.seed(42)
    lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=B, rand=True)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.25, hmc=False, network_arch=arch, num_steps=N,
                           eps_trainable=True)
    tr = GaugeTrainer(dyn, lr_init=1e-4)
    x = torch.as_tensor(lat.samples.reshape(B, -1), dtype=torch.float32, device="cuda")
    for _ in range(2):
        tr.train_step(x, 2.0)
    torch.cuda.synchronize()
    for name, fn in (("loss+grads", lambda: tr.calc_loss_and_grads(x, 2.0)), ("train_step", lambda: tr.train_step(x, 2.0))):
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        print(f"[{arch}] {name}: {dt*1e3:.2f} ms  ({2*B*N/dt/1e6:.2f} M chain-LF/s through forward+backward, rows={2*B})", flush=True)
    print("ws GB", la._lib.lib().l2hmc_gauge_train_ws_bytes(__import__('ctypes').byref(dyn._plan()), 2 * B) / 1e9)


if __name__ == "__main__":
    main()
