"""Device-resident sampling loop: the step harness around the hot path
(l2hmc/gauge_model.py:1304-1460 `GaugeModel.run`, the inference half of SURVEY.md 8f/f3).

The reference runs one `sess.run` per MCMC step, copying the whole sample batch host<->device through
`feed_dict` and wrapping it with `np.mod(., 2*pi)` on the host (:1371-1388).  Here the chain state never
leaves HBM: transition -> wrap -> per-step observables are all library calls on device buffers, and the
per-step scalars of all ranks are combined by one small asynchronous all-reduce (l2hmc_amd/dist.py)."""
import numpy as np
import torch

from . import _lib
from .dist import StepStats
from .lattice import u1_observables, u1_plaq_exact


class GaugeSampler:
    def __init__(self, dynamics, beta_init=2., beta_final=4., train_steps=10000, dist=None, reduce_every=16):
        self.dynamics = dynamics
        self.lattice = dynamics.lattice
        self.beta_init, self.beta_final, self.train_steps = beta_init, beta_final, train_steps
        # (reduce_every: MCMC steps whose scalar sums share one all-reduce when sharded; l2hmc_amd/dist.py)
        self.stats = StepStats(dynamics._device, dist, reduce_every=reduce_every)
        # [sum p, sum |dQ|, B, ticket] per step: the fused step kernel needs the ticket at 0 on entry and leaves it
        # at 0, so the rows of one zero-initialised ring are handed out in turn (longer than StepStats' backlog)
        self._sums_ring = torch.zeros(256, 4, dtype=torch.float32, device=dynamics._device)
        self._sums_next = 0

    def update_beta(self, step):
        """gauge_model.py:1039-1046: linear annealing of 1/beta."""
        temp = ((1. / self.beta_init - 1. / self.beta_final) * (1. - step / float(self.train_steps))
                + 1. / self.beta_final)
        return 1. / temp

    def wrap(self, x):
        out = torch.empty_like(x)
        _lib.check(_lib.lib().l2hmc_wrap_angle(x.data_ptr(), x.numel(), out.data_ptr(), _lib.stream_ptr(self.dynamics._device)))
        return out

    def step(self, x, beta):
        """One MCMC step on device state x: [B, x_dim].  Returns (x_next, px, observables of x, |dQ|);
        as in the reference the action / plaquette / charge ops look at the step's INPUT samples
        (gauge_model.py:256-266) and dQ compares input and output (:718-725).  Runs as ONE library call
        (l2hmc_gauge_mcmc_step: draws + trajectories + mix/accept + observables + wrap).  With
        `dynamics.both_directions = False` the plan carries L2HMC_PLAN_SELECTED_ONLY: each chain is integrated
        only in the direction its coin picks -- same random streams, same chains, half the work."""
        import ctypes as C
        dyn = self.dynamics
        x_next = torch.empty_like(x)                       # written by the step (out of place: no copy of x)
        B = x.shape[0]
        outs = {k: torch.empty(B, dtype=torch.float32, device=x.device)
                for k in ("px", "action", "avg_plaq", "top_charge", "dq")}
        sums = self._sums_ring[self._sums_next]                  # [sum p, sum |dQ|, B, ticket], filled in-kernel
        self._sums_next = (self._sums_next + 1) % self._sums_ring.shape[0]
        plan, L = dyn._plan(), _lib.lib()
        ws, nb = dyn._ws.get(L.l2hmc_gauge_mcmc_step_ws_bytes(C.byref(plan), B), x.device)
        # the step's random streams come from the dynamics' own draw counter (saved / restored with the state):
        # a sampler never replays noise the dynamics, a trainer or another sampler on the same dynamics used
        draw, dyn._draws = _lib.step_draw_index(dyn._draws)
        _lib.check(L.l2hmc_gauge_mcmc_step_ex(
            C.byref(plan), float(beta), _lib.dev_ptr(x, name="x"), x_next.data_ptr(), B, dyn._seed, draw,
            outs["px"].data_ptr(), outs["action"].data_ptr(), outs["avg_plaq"].data_ptr(),
            outs["top_charge"].data_ptr(), outs["dq"].data_ptr(), sums.data_ptr(), ws, nb,
            _lib.stream_ptr(dyn._device)))
        self.stats.push_sums(sums[:3])
        return x_next, outs["px"], outs, outs["dq"]

    def _step_composed(self, x, beta):
        """The same step from the separate public ops (used for selected-only mode and as a cross-check)."""
        T, X = self.lattice.time_size, self.lattice.space_size
        _, _, px, x_out = self.dynamics(x, beta)
        obs = u1_observables(x, T, X)
        x_next = self.wrap(x_out)
        dq = torch.abs(u1_observables(x_out, T, X)["top_charge"] - obs["top_charge"])
        self.stats.push(px, dq)
        return x_next, px, obs, dq

    METRICS = {'l1': 0, 'l2': 1, 'cos': 2, 'cos2': 3, 'cos_diff': 4}

    def calc_loss(self, x, beta, metric='cos_diff', loss_scale=1., z=None, **weights):
        """gauge_model.py:728-797 `_calc_loss`, forward value: (loss, x_out, px, x_dq).  Two transitions (on x
        and on the auxiliary z ~ N(0,1)), the per-chain terms in one kernel, and the mean over the chains of
        ALL ranks through one all-reduce of [sum, count] -- the collective north_star asks for."""
        if metric not in self.METRICS:        # :653-655
            raise AttributeError(f"metric={metric}. Expected one of: 'l1', 'l2', 'cos', 'cos2', or 'cos_diff'.")
        dyn = self.dynamics
        T, X = self.lattice.time_size, self.lattice.space_size
        x = _lib.as_dev(x, dyn._device)
        x_prop, _, px, x_out = dyn(x, beta)
        z = dyn._normal(tuple(x.shape)) if z is None else _lib.as_dev(z, dyn._device)
        _, _, pz, _ = dyn(z, beta)
        terms = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().l2hmc_gauge_loss_terms(
            x.data_ptr(), x_prop.data_ptr(), px.data_ptr(), z.data_ptr(), pz.data_ptr(), x.shape[0], T, X,
            self.METRICS[metric], float(loss_scale), float(weights.get('aux_weight', 1.)),
            float(weights.get('std_weight', 1.)), float(weights.get('charge_weight', 1.)), terms.data_ptr(),
            _lib.stream_ptr(self.dynamics._device)))
        buf = torch.stack([terms.sum(dtype=torch.float32),
                           torch.full((), float(terms.numel()), dtype=torch.float32, device=x.device)])
        if self.stats.dist is not None:
            self.stats.dist.all_reduce(buf, op=self.stats.dist.ReduceOp.SUM)
        loss = buf[0] / buf[1]
        q0 = u1_observables(x, T, X)["top_charge"]
        q1 = u1_observables(x_out, T, X)["top_charge"]
        x_dq = torch.abs(q0 - q1).to(torch.int32)          # :762-763
        self.last_loss_terms = terms
        return loss, x_out, px, x_dq

    def run(self, run_steps, beta, x=None, keep_samples=False):
        """:1304-1460 without the file/plot side effects.  Starts from N(0,1) samples like the reference
        (:1354) unless `x` is given.  Returns per-step histories as NumPy arrays [steps, B]."""
        dyn = self.dynamics
        if x is None:
            x = _lib.as_dev(np.random.randn(dyn.batch_size, dyn.x_dim), dyn._device)
        else:
            x = _lib.as_dev(x, dyn._device)
        hist = {k: [] for k in ("px", "actions", "plaqs", "charges", "charge_diff")}
        samples = []
        for _ in range(run_steps):
            x, px, obs, dq = self.step(x, beta)
            hist["px"].append(px)
            hist["actions"].append(obs["action"])
            hist["plaqs"].append(obs["avg_plaq"])
            hist["charges"].append(obs["top_charge"])
            hist["charge_diff"].append(dq)
            if keep_samples:
                samples.append(x)
        out = {k: torch.stack(v).cpu().numpy() for k, v in hist.items()}
        out["plaq_exact"] = u1_plaq_exact(beta)
        out["samples_out"] = x
        out["mean_accept"] = self.stats.mean_accept()
        if keep_samples:
            out["samples"] = torch.stack(samples).cpu().numpy()
        return out


    @staticmethod
    def run_dicts(out, beta):
        """`run()` histories -> the four dicts the reference pickles per run (gauge_model.py:1409-1413, :1758-1822:
        actions / plaqs / charges / charge_diff keyed by (step, beta)), for a caller that wants the reference's
        in-memory layout; this package itself writes .npz (save_run)."""
        n = out["px"].shape[0]
        keys = [(i, beta) for i in range(n)]
        return tuple({k: out[name][i] for i, k in enumerate(keys)}
                     for name in ("actions", "plaqs", "charges", "charge_diff"))

    def save_run(self, out, out_dir, beta, therm_frac=10):
        """gauge_model.py:1758-2033 (`_save_run_info`) without pickles: the histories returned by `run` go to
        `observables_steps_{n}_beta_{beta}.npz` and a readable summary -- per-chain means and standard errors of
        action, plaquette, topological charge and susceptibility after dropping the first steps // therm_frac
        steps, the charge histogram, mean accept probability, plaquette vs the exact value, integrated
        autocorrelation time of the plaquette -- to `statistics_steps_{n}_beta_{beta}.txt`.  Returns both paths."""
        import os
        from . import stats
        os.makedirs(out_dir, exist_ok=True)
        n = int(out["px"].shape[0])
        arrays = {k: np.asarray(v) for k, v in out.items()
                  if k in ("px", "actions", "plaqs", "charges", "charge_diff", "samples")}
        arrays["samples_out"] = out["samples_out"].detach().cpu().numpy()
        arrays["beta"], arrays["plaq_exact"] = np.float64(beta), np.float64(out["plaq_exact"])
        npz = os.path.join(out_dir, f"observables_steps_{n}_beta_{beta}.npz")
        np.savez_compressed(npz, **arrays)
        (am, ae), (pm, pe), (qm, qe), (sm, se), probs = stats.calc_observables_stats(
            out["actions"], out["plaqs"], out["charges"], therm_frac=therm_frac)
        therm = n // therm_frac
        try:
            tau = float(stats.integrated_time(out["plaqs"][therm:, :, None], quiet=True)[0][0])
        except Exception:      # noqa: BLE001 -- too short a run for the estimator
            tau = float("nan")
        lines = [f"run of {n} steps, {out['px'].shape[1]} chains, beta = {beta}, thermalisation cut {therm} steps",
                 f"mean accept probability      : {float(np.mean(out['px'][therm:])):.6f}",
                 f"average plaquette            : {float(pm.mean()):.6f} +/- {float(np.sqrt(np.mean(pe ** 2) / len(pe))):.6f}"
                 f"   (exact {float(out['plaq_exact']):.6f}, difference {float(pm.mean() - out['plaq_exact']):+.6f})",
                 f"average action               : {float(am.mean()):.6f} +/- {float(np.sqrt(np.mean(ae ** 2) / len(ae))):.6f}",
                 f"topological charge           : {float(qm.mean()):+.6f} +/- {float(np.sqrt(np.mean(qe ** 2) / len(qe))):.6f}",
                 f"topological susceptibility   : {float(sm.mean()):.6f} +/- {float(np.sqrt(np.mean(se ** 2) / len(se))):.6f}",
                 f"tunnelling rate |dQ| / step  : {float(np.mean(out['charge_diff'][therm:])):.6f}",
                 f"tau_int(plaquette)           : {tau:.3f} steps",
                 "charge probabilities         : " + ", ".join(f"{k:+d}: {v:.4f}" for k, v in probs.items())]
        txt = os.path.join(out_dir, f"statistics_steps_{n}_beta_{beta}.txt")
        with open(txt, "w") as f:
            f.write("\n".join(lines) + "\n")
        return npz, txt
