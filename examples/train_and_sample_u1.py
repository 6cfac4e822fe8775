"""End-to-end use of the library on the benchmark lattice: train the L2HMC sampler on 2D U(1) (8x8), then
sample with the trained networks and compare the average plaquette with the exact infinite-volume value.

    python examples/train_and_sample_u1.py [train_steps] [run_steps] [batch] [n_lf] [eps] [lr]

Mirrors what `python gauge_model.py --train_steps ... --run_steps ...` does in the reference (training loop
gauge_model.py:1119-1300, inference :1304-1460) without its file / plot side effects; everything between the
initial samples and the printed summary stays on the device."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import l2hmc_amd as la  # noqa: E402


def main():
    train_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    run_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    n_lf = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    eps0 = float(sys.argv[5]) if len(sys.argv) > 5 else 0.2
    # 3e-4: from a HOT start the reference's default 1e-3 (globals.py:31, meant for its cold start) survives or
    # diverges depending on the noise stream -- measured with two builds of the library, profiles/r02_example_*
    lr = float(sys.argv[6]) if len(sys.argv) > 6 else 3e-4
    L, beta = 8, 2.0
    np.random.seed(42)
    lat = la.GaugeLattice(L, L, 2, 'U1', num_samples=B, rand=True)
    dyn = la.GaugeDynamics(lat, lat.get_energy_function(), eps=eps0, hmc=False, network_arch='generic', num_steps=n_lf,
                           eps_trainable=True)
    sampler = la.GaugeSampler(dyn)
    x0 = torch.as_tensor(lat.samples.reshape(B, -1), dtype=torch.float32, device="cuda")

    def report(tag):
        out = sampler.__class__(dyn).run(run_steps, beta, x=x0)
        therm = run_steps // 5
        plaq = out["plaqs"][therm:]
        (_, _), (pm, pe), (_, _), (sm, _), probs = la.stats.calc_observables_stats(out["actions"], out["plaqs"],
                                                                                  out["charges"], therm_frac=5)
        tau, _ = la.stats.integrated_time(out["plaqs"][therm:, :, None], quiet=True)
        print(f"[{tag}] eps {float(dyn.eps):.4f}  mean accept {out['px'][therm:].mean():.3f}  "
              f"<plaq> {plaq.mean():.4f} +- {plaq.mean(1).std() / np.sqrt(len(plaq)):.4f}  (exact {out['plaq_exact']:.4f})  "
              f"tau_int(plaq) {tau[0]:.1f} steps  <Q^2> {sm.mean():.3f}  "
              f"tunnelling per step per chain {out['charge_diff'][therm:].mean():.4f}", flush=True)

    print(f"8x8 U(1), beta {beta}, {B} chains, {n_lf} LF steps, eps0 {eps0}, lr {lr}", flush=True)
    report("untrained")
    trainer = la.GaugeTrainer(dyn, lr_init=lr, lr_decay_steps=100, lr_decay_rate=0.96, clip_value=None)
    t0 = time.perf_counter()
    hist = trainer.train(train_steps, samples_init=x0, beta_init=beta, beta_final=beta, print_steps=max(1, train_steps // 10))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"trained {train_steps} steps in {dt:.2f} s ({1e3 * dt / train_steps:.2f} ms/step incl. logging); "
          f"loss {hist['loss'][0]:.1f} -> {hist['loss'][-1]:.1f}, accept {hist['accept_prob'][0]:.3f} -> "
          f"{hist['accept_prob'][-1]:.3f}", flush=True)
    report("trained")


if __name__ == "__main__":
    main()
