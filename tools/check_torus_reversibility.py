"""Evidence for a property of the REFERENCE algorithm (reproduced, not introduced, by this implementation):
gauge_model.py:1180,1388 wrap the samples to [0, 2 pi) between MCMC steps, but the S/T/Q networks take the raw
angles (generic_net.py:129-146) and the position update even rescales them (gauge_dynamics.py:519-531), so the
proposal map does not commute with 2 pi shifts: backward(forward(x)) returns to x, but
backward(wrap(forward(x))) does not.  The Markov chain on the torus is therefore not reversible, and a strongly
trained sampler converges to a biased plaquette.
    python tools/check_torus_reversibility.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import l2hmc_amd as la
np.random.seed(42)
B, n_lf = 2048, 10
lat = la.GaugeLattice(8, 8, 2, 'U1', num_samples=B, rand=True)
dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=0.25, hmc=False, network_arch='generic', num_steps=n_lf, eps_trainable=True)
x0 = torch.as_tensor(lat.samples.reshape(B, -1), dtype=torch.float32, device="cuda")

def torus_reversibility(tag, x):
    v = torch.randn_like(x)
    x1, v1, _ = dyn.transition_kernel(x, 2.0, forward=True, momentum=v)
    for wrap in (False, True):
        xin = torch.remainder(x1, 2 * np.pi) if wrap else x1
        x2, v2, _ = dyn.transition_kernel(xin, 2.0, forward=False, momentum=v1)
        d = torch.remainder(x2 - x + np.pi, 2 * np.pi) - np.pi          # angular distance
        print(f"[{tag}] backward(forward(x)){' with the intermediate state wrapped to [0, 2 pi)' if wrap else ''}: "
              f"rms angular distance to x = {float(d.pow(2).mean().sqrt()):.3e}, max {float(d.abs().max()):.3e}", flush=True)

torus_reversibility("untrained", x0)
tr = la.GaugeTrainer(dyn, lr_init=3e-4, lr_decay_steps=100, lr_decay_rate=0.96)
hist = tr.train(1000, samples_init=x0, beta_init=2.0, beta_final=2.0)
print("trained 1000 steps; accept", hist["accept_prob"][-1], flush=True)
torus_reversibility("trained", hist["samples"])
