"""Diagnostic (not shipped): per-phase cycle stamps of conv3d_front_kernel inside one sampling step of the layered
ConvNet3D path (stamped build, tools/build_diag.sh).    python tools/conv_fwd_stamps.py [chains] [L] [N_LF]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from l2hmc_amd import _lib
_lib.LIB_PATH = os.environ.get("L2HMC_DIAG_LIB", os.path.join(ROOT, "tools", "_diag", "libl2hmc_hip_diag.so"))
import l2hmc_amd as la
B, L, N = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 1024), (2, 16), (3, 15)))
np.random.seed(42)
lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=B, rand=True)
dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.2, hmc=False, network_arch='conv3D', num_steps=N,
                       eps_trainable=True, data_format='channels_last')
smp = la.GaugeSampler(dyn)
x = torch.as_tensor(lat.samples.reshape(B, -1), dtype=torch.float32, device="cuda")
for _ in range(3): x = smp.step(x, 2.0)[0]
torch.cuda.synchronize()
Lh = _lib.lib(); Lh.l2hmc_debug_set_stamps.argtypes = [C.c_void_p, C.c_int]
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device="cuda")
Lh.l2hmc_debug_set_stamps(stamps.data_ptr(), 7)
x = smp.step(x, 2.0)[0]; torch.cuda.synchronize()       # (every launch overwrites the slots: the step's LAST launch is read)
s = stamps.cpu().numpy().reshape(-1, 16); s = s[s[:, 0] != 0]
names = ["staging, halo zero", "barrier 1", "conv1 of group 0 (VALU alone)", "conv2 weight loads + barrier 2", "groups: conv2 MFMAs under conv1, last conv2, stores"]
tot = np.median(s[:, 5] - s[:, 0])
print(f"chains {B} lattice {L}x{L}: WGs {len(s)}  total cycles per WG (median) {tot:.0f}; grid span {(s[:,5].max()-s[:,0].min())} cycles")
for i, n in enumerate(names): print(f"  {n:38s} {np.median(s[:, i+1]-s[:, i]):9.0f} cyc  {100*np.median(s[:, i+1]-s[:, i])/tot:5.1f} %")
hw = s[:, 6] & 0xffffffff; xcc = (s[:, 6] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
simd = (hw >> 4) & 3
print("  SIMD of wave 0:", np.bincount(simd, minlength=4).tolist())
place = xcc * 4096 + se * 256 + sh * 16 + cu
u, cnt = np.unique(place, return_counts=True)
print(f"  distinct CUs used {len(u)}; workgroups per CU: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}; histogram {np.bincount(cnt).tolist()}")
xcc = (s[:, 6] >> 32) & 0xf
dur = (s[:, 15] - s[:, 7]) * 10.0   # ns (100 MHz)
print(f"  per-WG wall time median {np.median(dur)/1e3:.2f} us -> clock {np.median((s[:,5]-s[:,0]) / dur):.2f} GHz; first start to last end {(s[:,15].max()-s[:,7].min())/100:.2f} us")
print("  per-WG wall time percentiles (us):", [round(float(np.percentile(dur, q)) / 1e3, 2) for q in (0, 10, 50, 90, 99, 100)])
end = (s[:, 15] - s[:, 7].min()) / 100.0
print("  per-WG END time percentiles (us after first start):", [round(float(np.percentile(end, q)), 2) for q in (0, 10, 50, 90, 99, 100)])
print("  median wall per XCC:", [round(float(np.median(dur[xcc == k])) / 1e3, 2) for k in range(8)])
print("  p95 of the phases (cycles):", [int(np.percentile(s[:, i + 1] - s[:, i], 95)) for i in range(5)])
rt = s[:, 7] - s[:, 7].min()
print(f"  start times (100 MHz ticks): median {np.median(rt):.0f} p90 {np.percentile(rt, 90):.0f} max {rt.max()}")
Lh.l2hmc_debug_set_stamps(None, 0)
