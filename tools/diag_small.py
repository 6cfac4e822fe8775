"""Diagnostic (not shipped): where a wave of the toy-target trajectory kernel spends its cycles
(tools/build_diag.sh build, stamps of class 7).  cfg 2: 2-D mixture of Gaussians, 4096 chains, 10 LF, H = 50;
one `propose` = one launch, a wave walks 40 dependent network calls for its 8 chains x 2 directions."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from l2hmc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libl2hmc_hip_diag.so")
import l2hmc_amd as la  # noqa: E402
import bench  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    B = int(sys.argv[2]) if len(sys.argv) > 2 else bench.CONFIGS[cfg]["per_gpu"]
    N = bench.CONFIGS[cfg]["N"]
    L = _lib.lib()
    L.l2hmc_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
    dyn, target = bench.build_toy(cfg)
    x = torch.randn(B, 2, device="cuda")
    for _ in range(5):
        la.propose(x, dyn, do_mh_step=True)
    torch.cuda.synchronize()
    nw = (2 * B + 15) // 16
    stamps = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
    L.l2hmc_debug_set_stamps(stamps.data_ptr(), 7)
    for _ in range(3):
        stamps.zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        la.propose(x, dyn, do_mh_step=True)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    L.l2hmc_debug_set_stamps(None, 0)
    s = stamps.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 6] != 0]
    tot = np.median(s[:, 6])
    calls = 4 * N
    print(f"cfg {cfg}: {B} chains, {N} LF, {len(s)} waves; wall of one propose {wall * 1e6:.1f} us (stamped build); "
          f"median wave {tot:.0f} cycles = {tot / calls:.0f} per network call")
    names = ["first layer (matrix pipe, K = 2 dim + 4)", "hidden layer (MFMA 16x16x4) + bias + relu", "heads (MFMA) + bias",
             "ds_bpermute gather + tanh / exp(coeff)", "target energy + gradient", "sub-update arithmetic (exp, masks, log-det)"]
    acc = 0
    for i, n in enumerate(names):
        m = np.median(s[:, i])
        acc += m
        print(f"  {n:44s} {m:9.0f} cyc  {100 * m / tot:5.1f} %   {m / calls:7.0f} per call")
    print(f"  {'other (time encoding, loop, stamps)':44s} {tot - acc:9.0f} cyc  {100 * (tot - acc) / tot:5.1f} %")
    nt, ksh = {1: (1, 4), 2: (4, 14)}[cfg]          # latency form: first layer 2 x nt, hidden ksh x nt, heads ksh
    n_mfma = 2 * nt + ksh * nt + ksh
    ideal = calls * n_mfma * 32
    print(f"  MFMA issue cycles per wave ({n_mfma} instructions of 32 cycles per call): {ideal} = "
          f"{100 * ideal / tot:.1f} % of the wave's cycles")
    print("  (phase boundaries are taken at instruction ISSUE: a matrix instruction's completion is paid by the "
          "first phase that reads its result; the per-call total is the reliable figure)")


if __name__ == "__main__":
    main()
