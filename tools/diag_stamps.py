"""Diagnostic (not shipped, not a test): where do the dense kernels spend their
cycles?  Loads the -DL2HMC_STAMPS build, runs the S/T/Q network at the cfg-3
shape and prints, per kernel class, the median over workgroups of
  prologue (entry -> first tile in LDS), main loop, epilogue   [shader cycles]
and the in-kernel clock (cycles / s_memrealtime at 100 MHz)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libl2hmc_hip_diag.so")
import l2hmc_amd as la  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    T = X = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    arch = sys.argv[3] if len(sys.argv) > 3 else 'generic'
    D = 2 * T * X
    L = _lib.lib()
    L.l2hmc_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
    np.random.seed(106)
    if arch == 'generic':
        net = la.GenericNet(model_name='XNet', x_dim=D, num_hidden=4 * D, factor=2., name_scope='position',
                            links_shape=(T, X, 2))
    else:        # ConvNet3D as GaugeDynamics builds it (gauge_dynamics.py:121-143): F = T, H = 2 x_dim
        net = la.ConvNet3D(model_name='XNet', _input_shape=(rows, T, X, 2), links_shape=(T, X, 2), x_dim=D, factor=2.,
                           spatial_size=X, num_hidden=2 * D, num_filters=int(X), filter_sizes=[(3, 3, 2), (2, 2, 2)],
                           name_scope='position', data_format='channels_last')
    a = torch.randn(rows, D, device="cuda")
    b = torch.rand(rows, D, device="cuda") * 6.28
    t = np.array([[0.3, 0.95]])
    stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
    for _ in range(20):
        net([a, b, t])          # warm
    torch.cuda.synchronize()
    for cls, name in ((1, "L1 gemm_relu<64,1>"), (2, "L2 gemm_relu<64,2>"), (3, "heads")):
        L.l2hmc_debug_set_stamps(stamps.data_ptr(), cls)
        res = []
        for _ in range(10):
            stamps.zero_()
            net([a, b, t])
            torch.cuda.synchronize()
            s = stamps.cpu().numpy().reshape(-1, 8)
            s = s[s[:, 0] != 0]
            res.append(s)
        s = res[-1]
        pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
        tot = s[:, 3] - s[:, 0]
        real = (s[:, 5] - s[:, 4]) / 100e6       # seconds
        clk = tot / np.maximum(real, 1e-12) / 1e9
        span = (s[:, 5].max() - s[:, 4].min()) / 100e6 * 1e6
        start_skew = (s[:, 4] - s[:, 4].min()) / 100e6 * 1e6
        print(f"{name}: WGs {len(s)}  prologue {np.median(pro):.0f}  loop {np.median(loop):.0f}  epilogue {np.median(epi):.0f} "
              f"total {np.median(tot):.0f} cyc (max {tot.max()})  clock {np.median(clk):.2f} GHz  "
              f"per-WG {np.median(real)*1e6:.1f} us  grid span {span:.1f} us  start skew med {np.median(start_skew):.1f} max {start_skew.max():.1f} us",
              flush=True)
    L.l2hmc_debug_set_stamps(None, 0)


if __name__ == "__main__":
    main()
