"""Diagnostic: how long does the MCMC step take right after an idle period?  Per-step GPU time (events) of 80
consecutive steps following 2 s of idle, to size bench.py's untimed pre-warm (clock ramp)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from l2hmc_amd import GaugeSampler  # noqa: E402

dyn, *_ = bench.build_dynamics(bench.BATCH)
smp = GaugeSampler(dyn)
x = torch.rand(bench.BATCH, 128, device="cuda") * 6.28
for _ in range(3):
    x = smp.step(x, 2.0)[0]
torch.cuda.synchronize()
for trial in range(2):
    time.sleep(2.0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(81)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(80):
        x = smp.step(x, 2.0)[0]
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(80)]
    print(f"trial {trial}: wall {1e3 * wall / 80:.3f} ms/step; per-step GPU ms: first 10 "
          + " ".join(f"{m:.3f}" for m in ms[:10]) + f" | steps 10-19 mean {np.mean(ms[10:20]):.3f} | 20-39 {np.mean(ms[20:40]):.3f}"
          f" | 40-79 {np.mean(ms[40:]):.3f}")
