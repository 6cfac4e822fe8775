"""Diagnostic (not shipped): per-phase cycle stamps of conv3d_front_bwd_kernel inside one training step (stamped build).
    python tools/conv_bwd_stamps.py [chains] [L] [N_LF]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = "/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tools")
from l2hmc_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_diag", "libl2hmc_hip_diag.so")
import l2hmc_amd as la
from l2hmc_amd.gauge_trainer import GaugeTrainer
B, N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, int(sys.argv[3]) if len(sys.argv) > 3 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 8
np.random.seed(42)
lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=B, rand=True)
dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.2, hmc=False, network_arch='conv3D', num_steps=N,
                       eps_trainable=True, data_format='channels_last')
tr = GaugeTrainer(dyn, lr_init=1e-4)
x = torch.as_tensor(lat.samples.reshape(B, -1), dtype=torch.float32, device="cuda")
for _ in range(2): tr.train_step(x, 2.0)
Lh = _lib.lib(); Lh.l2hmc_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
Lh.l2hmc_debug_set_stamps(stamps.data_ptr(), 6)
tr.train_step(x, 2.0); torch.cuda.synchronize()
s = stamps.cpu().numpy().reshape(-1, 8); s = s[s[:, 0] != 0]
names = ["setup+stage", "phase1 conv1 fwd", "phase2 conv2 fwd", "phase3 dpool1", "phase4 dinput", "phase5 filter grads", "phase5b slot add"]
tot = np.median(s[:, 7] - s[:, 0])
print("WGs", len(s), "total cycles (median)", tot)
for i, n in enumerate(names): print(f"  {n:22s} {np.median(s[:, i+1]-s[:, i]):9.0f} cyc  {100*np.median(s[:, i+1]-s[:, i])/tot:5.1f} %")
Lh.l2hmc_debug_set_stamps(None, 0)
