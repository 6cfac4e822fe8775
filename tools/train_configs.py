"""Training step (loss + reverse pass + Adam; gauge_model.py:728-832, :925-970) at the shapes of BASELINE.json
configs[2..4], per-GPU batch: time per step, workspace (tape) bytes, executed-FLOP rate.
    python tools/train_configs.py [cfg ...]        (default: 3 4 5)
A training step integrates every chain and every auxiliary chain in the direction its coin selects: 2 B rows.
FLOPs per step: forward 8 MACs-per-call per row per LF step (4 calls x 2), reverse pass twice that
(backward-data + weight-gradient products) => 3 x forward."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from l2hmc_amd import _lib  # noqa: E402
from l2hmc_amd.gauge_trainer import GaugeTrainer  # noqa: E402


def run(cfg, B, iters):
    c = bench.CONFIGS[cfg]
    dyn = bench.build_gauge(cfg, B)
    D = 2 * c["L"] ** 2
    x = torch.rand(B, D, device="cuda") * (2 * np.pi)
    ws = _lib.lib().l2hmc_gauge_train_ws_bytes(C.byref(dyn._plan()), 2 * B)
    print(f"cfg {cfg}: {c['L']}x{c['L']} {c['arch']}, {B} chains + {B} auxiliary chains, {c['N']} LF; "
          f"training workspace (tape of every network call + deltas) {ws / 1e9:.2f} GB", flush=True)
    tr = GaugeTrainer(dyn, lr_init=1e-5)
    loss = tr.train_step(x, c["beta"])[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss, x_out, px, _ = tr.train_step(x, c["beta"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    macs, _ = bench.config_macs(cfg)
    flops = 3 * 8 * macs * 2 * B * c["N"]
    g = tr.grads
    print(f"   {dt * 1e3:10.2f} ms / training step   {flops / dt / 1e12:6.1f} TFLOP/s ({flops / dt / 1e12 / 157.3:.2f} of the fp32 "
          f"MFMA peak)   loss {float(loss):.4f}  mean accept {float(px.mean()):.4f}  |grad| finite: "
          f"{bool(torch.isfinite(g).all())}  max |grad| {float(g.abs().max()):.3e}   peak device memory "
          f"{torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
    del tr, dyn, x
    torch.cuda.empty_cache()


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [3, 4, 5]
    print("device", torch.cuda.get_device_name(0), "| library", _lib.LIB_PATH, flush=True)
    for cfg in cfgs:
        B = bench.CONFIGS[cfg]["per_gpu"]
        run(cfg, B, {3: 10, 4: 3, 5: 1}[cfg])


if __name__ == "__main__":
    main()
