"""Batch sweep of the whole-step kernel (K1) at the benchmark dynamics (8x8, beta 2, 10 LF, GenericNet H=512):
ms per MCMC step and useful chain-LF/s against the number of chains on ONE GPU.  16 chain-rows per workgroup and
one workgroup per CU make the kernel a staircase in rows / (16 x 256); this table is what shows it.
    python tools/batch_sweep.py [--arch generic|conv3D] [--selected-only] > gpurun_out/batch_sweep.txt
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from l2hmc_amd import GaugeSampler, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="generic")
ap.add_argument("--selected-only", action="store_true")
ap.add_argument("--batches", default="256,512,1024,1536,2040,2047,2048,2049,2056,2304,2560,3072,4096,4097,6144,8192,16384")
args = ap.parse_args()

print(f"# device {torch.cuda.get_device_name(0)}; library {_lib.LIB_PATH}")
print(f"# cfg 3 dynamics, arch {args.arch}, {'selected direction only' if args.selected_only else 'both directions'}")
print("# chains  rows  workgroups(16 rows)  rounds(256 CUs)   ms/step   useful M chain-LF/s   TFLOP/s executed  frac of 157.3")
macs = bench.net_macs(128, 512) if args.arch == "generic" else bench.conv_front_macs(8, 8) + bench.net_macs(128, 256, 64, 64)
prev = 0.0
for B in [int(b) for b in args.batches.split(",")]:
    dyn = bench.build_gauge(3, B, not args.selected_only, arch=args.arch)
    smp = GaugeSampler(dyn)
    x = torch.rand(B, 128, device="cuda") * (2 * np.pi)
    for _ in range(60):
        x = smp.step(x, 2.0)[0]
    torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter()
    for _ in range(n):
        x = smp.step(x, 2.0)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    rows = B * (1 if args.selected_only else 2)
    wgs = -(-rows // 16)
    rate = B * 10 / dt
    tf = rows * 10 * 8 * macs / dt / 1e12
    flag = "" if rate >= prev else "   <-- NOT monotone"
    prev = max(prev, rate)
    print(f"{B:7d} {rows:6d} {wgs:8d} {wgs / 256:14.2f} {dt * 1e3:12.3f} {rate / 1e6:14.3f} {tf:16.1f} {tf / 157.3:10.3f}{flag}",
          flush=True)
    del dyn, smp, x
