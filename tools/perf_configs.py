"""Diagnostic (not a test, not the bench contract): per-GPU timings of every BASELINE.json config
through the public host classes, plus the HBM roofline of the standalone U(1) kernel.
    python tools/perf_configs.py > gpurun_out/perf_configs.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import l2hmc_amd as la  # noqa: E402
if os.environ.get("L2HMC_LIB"):          # a diagnostic build (e.g. the previous commit's library, for before/after lines)
    from l2hmc_amd import _lib as _l
    _l.LIB_PATH = os.path.abspath(os.environ["L2HMC_LIB"])


def timeit(fn, warm=2, iters=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def generic_macs(D, H):
    return 2 * D * H + 2 * H + H * H + 3 * H * D


def conv_macs(L, D, H, F):
    return 2 * (L * L * 2 * F * 18 + (L // 2) ** 2 * 2 * F * 8 * F + (L // 4) ** 2 * 2 * F * H) + 2 * H + H * H + 3 * H * D


def gauge(name, L, B, N, eps, beta, arch, iters=5, fused=True):
    np.random.seed(42)
    lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=B, rand=True)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=eps, hmc=False, network_arch=arch, num_steps=N,
                           eps_trainable=True, data_format='channels_last')
    dyn.fused = fused
    x = torch.as_tensor(lat.samples.reshape(B, -1), dtype=torch.float32, device="cuda")
    dt = timeit(lambda: dyn(x, beta), warm=4, iters=iters)
    D = 2 * L * L
    macs = generic_macs(D, 4 * D) if arch == 'generic' else conv_macs(L, D, 2 * D, L)
    flops = 2 * B * N * 8 * macs
    print(f"{name:46s} {dt*1e3:10.3f} ms/transition  useful {B*N/dt/1e6:8.3f} M chain-LF/s  "
          f"{flops/dt/1e12:6.1f} TFLOP/s executed ({flops/dt/1e12/157.3:.2f} of fp32 MFMA peak)", flush=True)
    del dyn, lat
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    time.sleep(0.3)       # the release of the big layered workspaces otherwise runs into the next configuration's timing


FIRST_LAYER_FORM = 0


def small(name, target, B, N, H):
    dyn = la.Dynamics(2, target.get_energy_function(), trajectory_length=N, eps=0.1,
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=H))
    dyn.first_layer_form = FIRST_LAYER_FORM
    x = torch.randn(B, 2, device="cuda")
    dt = timeit(lambda: la.propose(x, dyn, do_mh_step=True), warm=3, iters=20)
    print(f"{name:46s} {dt*1e3:10.3f} ms/propose     useful {B*N/dt/1e6:8.3f} M chain-LF/s", flush=True)


def small_train(name, target, B, N, H):
    from l2hmc_amd.dynamics_trainer import DynamicsTrainer
    dyn = la.Dynamics(2, target.get_energy_function(), trajectory_length=N, eps=0.1,
                      net_factory=lambda d, scope, factor: la.network(d, scope, factor, num_nodes=H))
    tr = DynamicsTrainer(dyn, lr_init=1e-3)
    x = torch.randn(B, 2, device="cuda")
    dt = timeit(lambda: tr.train_step(x), warm=3, iters=20)
    print(f"{name:46s} {dt*1e3:10.3f} ms/train step  ({2*B} chains: forward + loss + reverse pass + Adam)", flush=True)


def u1_roofline():
    """K3 standalone: HBM-bound stencil, 8*D bytes per chain (+ 4 per scalar observable)."""
    from l2hmc_amd import _lib
    print("library:", _lib.LIB_PATH)
    for L, rows in ((8, 1 << 21), (16, 1 << 19), (32, 1 << 17)):
        D = 2 * L * L
        x = torch.rand(rows, D, device="cuda") * 6.28
        f = torch.empty_like(x)
        a, pl, q = (torch.empty(rows, device="cuda") for _ in range(3))
        Lh = _lib.lib()
        for what, ptrs, extra in (("force only (the integrator's call)", (None, f.data_ptr(), None, None), 0),
                                  ("action + force", (a.data_ptr(), f.data_ptr(), None, None), 4),
                                  ("action + plaquette + charge + force", (a.data_ptr(), f.data_ptr(), pl.data_ptr(),
                                                                            q.data_ptr()), 12)):
            def run():
                _lib.check(Lh.l2hmc_u1_action_force(x.data_ptr(), rows, L, L, 2.0, *ptrs, _lib.stream_ptr()))
            dt = timeit(run, warm=3, iters=10)
            nbytes = rows * (8 * D + extra)
            print(f"u1_action_force {L}x{L} rows={rows} {what:38s}: {dt*1e3:.3f} ms  {nbytes/dt/1e9:.0f} GB/s algorithmic "
                  f"({nbytes/dt/8e12:.2f} of 8 TB/s)", flush=True)
        # the same buffers through a plain device copy and an element-wise op, timed the same way: what this
        # read-once / write-once pattern reaches on this part without any arithmetic or neighbour exchange
        for what, fn in (("torch copy_ (same bytes: reference point)", lambda: f.copy_(x)),
                         ("torch sin(x, out=f) (same bytes: reference point)", lambda: torch.sin(x, out=f))):
            dt = timeit(fn, warm=3, iters=10)
            nbytes = rows * 8 * D
            print(f"   {what:60s}: {dt*1e3:.3f} ms  {nbytes/dt/1e9:.0f} GB/s ({nbytes/dt/8e12:.2f} of 8 TB/s)", flush=True)
        del x, f


def main():
    print("device", torch.cuda.get_device_name(0), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "u1":
        u1_roofline()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "small":
        from l2hmc_amd import _lib
        print("library:", _lib.LIB_PATH)
        if len(sys.argv) > 2:               # 1: first layer on the matrix pipe, 2: on the VALU (default: by batch size)
            global FIRST_LAYER_FORM
            FIRST_LAYER_FORM = int(sys.argv[2])
            print("first-layer form forced to", {1: "matrix pipe", 2: "VALU"}[int(sys.argv[2])])
        scg = la.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]]))
        mog = la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5])
        small("cfg1 SCG 2-D, B=128, 5 LF, H=10", scg, 128, 5, 10)
        small("cfg2 MoG 2-D, B=4096, 10 LF, H=50", mog, 4096, 10, 50)
        small("     MoG 2-D, B=65536, 10 LF, H=50 (chip-filling batch)", mog, 65536, 10, 50)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cfg4":
        gauge("cfg4 U(1) 16x16 conv3D, B=1024/GPU, 15 LF", 16, 1024, 15, 0.2, 3.0, 'conv3D', iters=3)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cfg4g":
        gauge("cfg4-like U(1) 16x16 generic, B=1024/GPU, 15 LF", 16, 1024, 15, 0.2, 3.0, 'generic', iters=3)
        return
    u1_roofline()
    small("cfg1 SCG 2-D, B=128, 5 LF, H=10",
          la.Gaussian(np.zeros(2), np.array([[50.05, -49.95], [-49.95, 50.05]])), 128, 5, 10)
    small("cfg2 MoG 2-D, B=4096, 10 LF, H=50",
          la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5]), 4096, 10, 50)
    small_train("cfg2 MoG training step, B=4096, 10 LF, H=50",
                la.GMM([np.array([1., 0.]), np.array([0., 1.])], [0.025 * np.eye(2)] * 2, [0.5, 0.5]), 4096, 10, 50)
    gauge("cfg3 U(1) 8x8 generic, B=2048, 10 LF (fused)", 8, 2048, 10, 0.25, 2.0, 'generic', iters=10)
    gauge("cfg3 U(1) 8x8 generic, B=2048, 10 LF (layered)", 8, 2048, 10, 0.25, 2.0, 'generic', iters=10, fused=False)
    gauge("cfg3 U(1) 8x8 conv3D,  B=2048, 10 LF", 8, 2048, 10, 0.25, 2.0, 'conv3D', iters=10)
    gauge("cfg4 U(1) 16x16 conv3D, B=1024/GPU, 15 LF", 16, 1024, 15, 0.2, 3.0, 'conv3D', iters=3)
    gauge("cfg5 U(1) 32x32 generic, B=2048/GPU, 25 LF", 32, 2048, 25, 0.1, 4.0, 'generic', iters=2)


if __name__ == "__main__":
    main()
