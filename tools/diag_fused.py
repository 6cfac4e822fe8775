"""Diagnostic (not shipped): cycle shares inside the fused trajectory kernel (wave 0 of every workgroup)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import _lib  # noqa: E402

NW = os.environ.get("DIAG_WAVES", "4")
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", f"libl2hmc_hip_diag_w{NW}.so")
import l2hmc_amd as la  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    arch = sys.argv[2] if len(sys.argv) > 2 else 'generic'
    T = X = 8
    N, eps, beta = 10, 0.25, 2.0
    L = _lib.lib()
    L.l2hmc_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
    np.random.seed(106)
    lat = la.GaugeLattice(T, X, 2, 'U1', num_samples=B, rand=False)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=eps, hmc=False, network_arch=arch, num_steps=N,
                           eps_trainable=True, data_format='channels_last')
    x = torch.rand(B, 128, device="cuda") * 6.28
    vf, vb = torch.randn(B, 128, device="cuda"), torch.randn(B, 128, device="cuda")
    coin, u = torch.rand(B, device="cuda"), torch.rand(B, device="cuda")
    run = lambda: dyn.apply_transition(x, beta, vf, vb, coin, u)      # noqa: E731 -- the two trajectories alone
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    stamps = torch.zeros(4096 * 12, dtype=torch.int64, device="cuda")
    L.l2hmc_debug_set_stamps(stamps.data_ptr(), 5)
    stag = int(os.environ.get("DIAG_STAGGER", "0"))
    L.l2hmc_debug_set_stagger.argtypes = [C.c_int]
    L.l2hmc_debug_set_stagger(stag)
    print(f"--- waves/WG {NW}  stagger {stag} cycles/WG")
    import time
    for it in range(3):
        stamps.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    print(f"wall of one transition {wall*1e3:.3f} ms")
    s = stamps.cpu().numpy().reshape(-1, 12)
    s = s[s[:, 9] != 0]
    names = ["gemm L1", "gemm L2", "gemm heads", "epi L1", "epi L2", "epi heads", "barriers", "force",
             "conv stage (inside gemm L1)" if arch != 'generic' else "mask pass"]
    tot = np.median(s[:, 9])
    real = (s[:, 11] - s[:, 10]) / 100e6
    print(f"WGs {len(s)}  total {tot:.0f} cyc  per-WG {np.median(real)*1e3:.3f} ms  clock {np.median(s[:,9]/real)/1e9:.2f} GHz  "
          f"grid span {(s[:,11].max()-s[:,10].min())/100e6*1e3:.3f} ms")
    acc = 0
    for i, n in enumerate(names):
        m = np.median(s[:, i])
        acc += m if i != 8 or arch == 'generic' else 0
        print(f"  {n:12s} {m:10.0f} cyc  {100*m/tot:5.1f} %")
    print(f"  {'other':12s} {tot-acc:10.0f} cyc  {100*(tot-acc)/tot:5.1f} %")
    ideal = 10 * 4 * (32 + 32 + 32 * 6 / 8) * 32 * 32 / 1  # per wave: kchunks*tiles*4 MFMAs*32 cyc
    # per net call per wave: L1 16 chunks*8 tiles*4 = 512 MFMA; L2 32*8*4 = 1024; heads 32*6*4 = 768 -> 2304 MFMAs * 32 cyc
    print(f"  ideal MFMA issue cycles per wave: {40*2304*32} ({100*40*2304*32/tot:.1f} % of total)")
    L.l2hmc_debug_set_stamps(None, 0)


if __name__ == "__main__":
    main()
