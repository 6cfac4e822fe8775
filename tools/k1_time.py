"""Quick A/B timing of the whole-step kernel at the headline shape (not shipped): ms per MCMC step of GaugeSampler.step
over 2048 / 4096 chains (16-row and 32-row forms) and 512 chains (sub-tile form), library chosen by L2HMC_LIB_PATH."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import l2hmc_amd as la  # noqa: E402
from l2hmc_amd import _lib  # noqa: E402


def main():
    print("library:", _lib.LIB_PATH, flush=True)
    out = []
    for B in [int(a) for a in sys.argv[1:]] or [2048, 4096, 512]:
        np.random.seed(106)
        lat = la.GaugeLattice(8, 8, 2, 'U1', num_samples=B, rand=False)
        dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.25, hmc=False, network_arch='generic', num_steps=10,
                               eps_trainable=True, data_format='channels_last')
        smp = la.GaugeSampler(dyn)
        x = torch.rand(B, 128, device="cuda") * 6.28
        for _ in range(150):
            x = smp.step(x, 2.0)[0]
        smp.stats.wait()
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                x = smp.step(x, 2.0)[0]
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 100)
            smp.stats.wait()
        out.append(f"{B} chains {best*1e3:.4f} ms")
    print("  ".join(out), flush=True)


if __name__ == "__main__":
    main()
