"""Training step of the lattice sampler on the device:
  l2hmc/gauge_model.py:728-830  _calc_loss / _calc_loss_and_grads (loss + tf.gradients + clip_by_global_norm)
  l2hmc/gauge_model.py:925-970  exponential_decay schedule, AdamOptimizer (x hvd.size()), apply_gradients
  l2hmc/gauge_model.py:1160-1200 one `train_op` evaluation per step.

Everything between "samples in" and "weights updated" is library calls on device buffers
(include/l2hmc_hip.h, training section); the only host work per step is reading back the new step size,
which the plan passes by value.  Gradients of all ranks are combined by ONE all-reduce(SUM) of a flat buffer
[xnet | vnet | eps] -- the loss already divides by the global chain count, so the sum IS the gradient of the
global mean.  (The reference calls tf.gradients + apply_gradients, which bypasses
hvd.DistributedOptimizer.compute_gradients: its ranks never exchange gradients.  `allreduce_grads=False`
reproduces that; the default does what the Horovod wrapper is there for.)

Chains are integrated only in the direction their coin selects: the reference's mask multiplies the other
direction's outputs by exactly 0, so it contributes exactly 0 to every gradient."""
import ctypes as C

import torch

from . import _lib
from .dist import active as _active_dist
from .lattice import u1_observables

METRICS = {'l1': 0, 'l2': 1, 'cos': 2, 'cos2': 3, 'cos_diff': 4}


class GaugeTrainer:
    def __init__(self, dynamics, lr_init=1e-3, lr_decay_steps=1000, lr_decay_rate=0.96, clip_value=None,
                 metric='cos_diff', loss_scale=1., aux_weight=1., std_weight=1., charge_weight=1., dist=None,
                 allreduce_grads=True, beta1=0.9, beta2=0.999, epsilon=1e-8, eager_variables=False,
                 bucketed=True):
        if metric not in METRICS:        # gauge_model.py:653-655
            raise AttributeError(f"metric={metric}. Expected one of: 'l1', 'l2', 'cos', 'cos2', or 'cos_diff'.")
        if dynamics.hmc:
            raise ValueError("hmc=True dynamics have no trainable networks (gauge_model.py:913-918 builds the "
                             "sampler only)")
        if dynamics.network_arch not in ('generic', 'conv3D'):
            raise NotImplementedError("training is implemented for network_arch 'generic' and 'conv3D'")
        self.dynamics = dyn = dynamics
        self.metric, self.loss_scale = metric, float(loss_scale)
        self.weights = dict(aux_weight=float(aux_weight), std_weight=float(std_weight),
                            charge_weight=float(charge_weight))
        self.lr_init, self.lr_decay_steps, self.lr_decay_rate = float(lr_init), int(lr_decay_steps), float(lr_decay_rate)
        self.clip_value = None if clip_value is None else float(clip_value)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.dist = _active_dist(dist)
        self.world = self.dist.get_world_size() if self.dist is not None else 1
        self.allreduce_grads = bool(allreduce_grads)
        # bucketed: gradient groups go on the wire as soon as the reverse pass has produced them, on a side stream,
        # while the next group's products still run (what hvd.DistributedOptimizer does tensor by tensor,
        # gauge_model.py:942-943); False = one all-reduce of the whole buffer after the pass.  Same numbers.
        self.bucketed = bool(bucketed)
        # Which variable list the optimiser sees.  Graph mode -- the path the CLI runs -- differentiates and
        # applies over `dynamics.variables` (gauge_model.py:825, :965-968), which holds the step size even when
        # it was created with trainable=False (gauge_dynamics.py:91-96): eps moves and counts in the clip norm
        # REGARDLESS of `eps_trainable`.  Eager mode uses `trainable_variables` (:820) and respects the flag.
        # Default = graph mode, as the reference's sessions run; eager_variables=True = the eager branch.
        self.eager_variables = bool(eager_variables)
        self.global_step = 0
        self.last_bucket_count = 0       # gradient groups the last step put on the wire (0: nothing to exchange)
        dev = dyn._device
        self._nets = (dyn.position_fn, dyn.momentum_fn)
        flats = [n.flat_params() for n in self._nets]
        self._sizes = [f[0].numel() for f in flats]
        n_all = sum(self._sizes) + 1
        # one bucket: [xnet | vnet | eps]
        self.grads = torch.zeros(n_all, dtype=torch.float32, device=dev)
        self._m = torch.zeros_like(self.grads)
        self._v = torch.zeros_like(self.grads)
        self._eps_dev = torch.tensor([float(dyn.eps)], dtype=torch.float32, device=dev)
        self._gnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self._ws = _lib.Workspace()
        self.broadcast_weights()
        self._grad_structs, self._conv_grad_structs = [], []
        off = 0
        for net, (flat, views, offsets) in zip(self._nets, flats):
            at = lambda k: self.grads.data_ptr() + 4 * (off + offsets[k][0])     # noqa: E731
            self._grad_structs.append(_lib.DenseGrads(**{k: at(k) for k in net.SEGMENTS}))
            conv = [k for k in offsets if k not in net.SEGMENTS]
            self._conv_grad_structs.append(_lib.Conv3DGrads(**{k: at(k) for k in conv}) if conv else None)
            off += flat.numel()
        self._buckets = self._bucket_ranges()
        self._side = torch.cuda.Stream(device=dev) if (self.dist is not None and dev.type == "cuda") else None

    def _bucket_ranges(self):
        """{bucket id of l2hmc_gauge_train_backward_buckets: [(lo, hi), ...]} over the flat gradient buffer
        [xnet | vnet | eps]; a network's flat layout is SEGMENTS (+ Conv3D kernels), so groups 3k .. 3k+2 are
        single contiguous ranges and the rest (front-end gradients, eps) a few short ones."""
        out, rest, off = {}, [], 0
        for k, net in enumerate(self._nets):
            flat, _, offsets = net.flat_params()
            out[3 * k + 0] = [(off + offsets["w1_t"][0], off + offsets["b1"][1])]
            out[3 * k + 1] = [(off + offsets["wh_t"][0], off + offsets["bh"][1])]
            out[3 * k + 2] = [(off + offsets["whd_t"][0], off + offsets["coeff_q"][1])]
            extra = [n for n in offsets if n not in net.SEGMENTS]
            if extra:
                rest.append((off + min(offsets[n][0] for n in extra), off + max(offsets[n][1] for n in extra)))
            off += flat.numel()
        rest.append((off, off + 1))
        out[_lib.GRAD_BUCKET_REST] = rest
        return out

    def broadcast_weights(self, src=0):
        """hvd.BroadcastGlobalVariablesHook(0) (gauge_model.py:1008): every rank starts from rank `src`'s
        weights, step size and masks -- one broadcast per flat buffer."""
        if self.dist is None:
            return
        dyn = self.dynamics
        for net in self._nets:
            self.dist.broadcast(net.flat_params()[0], src=src)
            net.refresh_packed()
        self.dist.broadcast(self._eps_dev, src=src)
        dyn.eps = self._eps_dev.detach().cpu().reshape(())
        self.dist.broadcast(dyn.mask, src=src)

    # ---- views ----------------------------------------------------------------
    def grad_views(self):
        """{'xnet': {segment: tensor}, 'vnet': {...}, 'eps': tensor} over the flat gradient bucket."""
        out, off = {}, 0
        for name, net in zip(("xnet", "vnet"), self._nets):
            flat, views, offsets = net.flat_params()
            out[name] = {k: self.grads[off + a:off + b].view(views[k].shape) for k, (a, b) in offsets.items()}
            off += flat.numel()
        out["eps"] = self.grads[off:off + 1]
        return out

    def learning_rate(self):
        """tf.train.exponential_decay(staircase=True) times the rank count (gauge_model.py:934-942)."""
        return self.lr_init * self.lr_decay_rate ** (self.global_step // self.lr_decay_steps) * self.world

    # ---- loss + gradients ----------------------------------------------------------
    def calc_loss_and_grads(self, x, beta, z=None, draws_x=None, draws_z=None):
        """gauge_model.py:799-830 -> (loss, x_out, accept_prob, x_dq); gradients land in self.grads.
        draws_*: optional (momentum_f, momentum_b, coin, u) for the x and the z transition."""
        dyn = self.dynamics
        dev = dyn._device
        T, X = dyn.lattice.time_size, dyn.lattice.space_size
        x = dyn._x(x)
        B, D = x.shape
        z = dyn._normal((B, D)) if z is None else dyn._x(z)

        def draw(d):
            if d is None:
                return dyn._normal((B, D)), dyn._normal((B, D)), dyn._uniform((B,)), dyn._uniform((B,))
            return tuple(_lib.as_dev(a, dev) for a in d)
        vf_x, vb_x, coin_x, u_x = draw(draws_x)
        vf_z, vb_z, coin_z, _ = draw(draws_z)
        coin = torch.cat([coin_x, coin_z])
        fwd = coin > 0.5
        x0 = torch.cat([x, z]).contiguous()
        v0 = torch.where(fwd[:, None], torch.cat([vf_x, vf_z]), torch.cat([vb_x, vb_z])).contiguous()
        dirs = (~fwd).to(torch.int32).contiguous()
        R = 2 * B
        xN, vN = torch.empty_like(x0), torch.empty_like(x0)
        sld, p = (torch.empty(R, dtype=torch.float32, device=dev) for _ in range(2))
        plan, L = dyn._plan(), _lib.lib()
        ws, nb = self._ws.get(L.l2hmc_gauge_train_ws_bytes(C.byref(plan), R), dev)
        s = _lib.stream_ptr(self.dynamics._device)
        _lib.check(L.l2hmc_gauge_train_forward(C.byref(plan), float(beta), x0.data_ptr(), v0.data_ptr(),
                                               dirs.data_ptr(), R, xN.data_ptr(), vN.data_ptr(), sld.data_ptr(),
                                               p.data_ptr(), ws, nb, s))
        terms = torch.empty(B, dtype=torch.float32, device=dev)
        dxN, dvN = torch.empty_like(x0), torch.empty_like(x0)
        dld = torch.empty(R, dtype=torch.float32, device=dev)
        w = self.weights
        _lib.check(L.l2hmc_gauge_loss_backward(T, X, float(beta), x0.data_ptr(), xN.data_ptr(), vN.data_ptr(),
                                               p.data_ptr(), B, METRICS[self.metric], self.loss_scale,
                                               w['aux_weight'], w['std_weight'], w['charge_weight'],
                                               1.0 / (B * self.world), terms.data_ptr(), dxN.data_ptr(),
                                               dvN.data_ptr(), dld.data_ptr(), s))
        n0, n1 = self._sizes
        bargs = (C.byref(plan), float(beta), dirs.data_ptr(), R, dxN.data_ptr(), dvN.data_ptr(), dld.data_ptr(),
                 C.byref(self._grad_structs[0]), C.byref(self._grad_structs[1]),
                 *(C.byref(g) if g is not None else None for g in self._conv_grad_structs),
                 self.grads.data_ptr() + 4 * (n0 + n1), ws, nb, s)
        buf = torch.stack([terms.sum(dtype=torch.float32),
                           torch.full((), float(B), dtype=torch.float32, device=dev)])
        if self.dist is not None and self.allreduce_grads and self.bucketed:
            works, errors = [], []
            cur = torch.cuda.current_stream(dev) if dev.type == "cuda" else None

            def on_bucket(_user, b):           # host thread, right after bucket b's producers were enqueued
                try:
                    if self._side is not None:
                        self._side.wait_stream(cur)
                        with torch.cuda.stream(self._side):
                            for lo, hi in self._buckets[int(b)]:
                                works.append(self.dist.all_reduce(self.grads[lo:hi], op=self.dist.ReduceOp.SUM,
                                                                  async_op=True))
                    else:
                        for lo, hi in self._buckets[int(b)]:
                            works.append(self.dist.all_reduce(self.grads[lo:hi], op=self.dist.ReduceOp.SUM,
                                                              async_op=True))
                except Exception as e:          # noqa: BLE001 -- a ctypes callback cannot raise: re-raised below
                    errors.append(e)
            cb = _lib.BUCKET_FN(on_bucket)
            _lib.check(L.l2hmc_gauge_train_backward_buckets(*bargs, cb, None))
            if errors:
                raise errors[0]
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
            for wk in works:
                wk.wait()
            if self._side is not None:
                cur.wait_stream(self._side)
            self.last_bucket_count = len(works)
        else:
            _lib.check(L.l2hmc_gauge_train_backward(*bargs))
            if self.dist is not None:
                self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
                if self.allreduce_grads:
                    self.dist.all_reduce(self.grads, op=self.dist.ReduceOp.SUM)
        loss = buf[0] / buf[1]
        px = p[:B]
        x_prop = xN[:B]
        x_out = torch.where((px > u_x)[:, None], x_prop, x)        # gauge_dynamics.py:244-257 (strict >)
        q0 = u1_observables(x, T, X)["top_charge"]
        q1 = u1_observables(x_out, T, X)["top_charge"]
        x_dq = torch.abs(q0 - q1).to(torch.int32)                  # gauge_model.py:762-763
        self.last_loss_terms, self.last_pz = terms, p[B:]
        return loss, x_out, px, x_dq

    # ---- optimiser --------------------------------------------------------------
    def apply_gradients(self):
        """clip_by_global_norm (if clip_value) + Adam on [xnet | vnet | eps]; bumps global_step."""
        dyn, L, s = self.dynamics, _lib.lib(), _lib.stream_ptr(self.dynamics._device)
        lr = self.learning_rate()
        self._adam_t = getattr(self, "_adam_t", 0) + 1
        t = self._adam_t
        lr_t = lr * (1. - self.beta2 ** t) ** 0.5 / (1. - self.beta1 ** t)
        n0, n1 = self._sizes
        gp, mp, vp = self.grads.data_ptr(), self._m.data_ptr(), self._v.data_ptr()
        segs = []
        off = 0
        for net in self._nets:
            flat, _, offsets = net.flat_params()
            segs.append((flat.data_ptr(), off, flat.numel(), offsets["b1"]))
            off += flat.numel()
        trainable_eps = bool(dyn.eps_trainable) or not self.eager_variables
        gnorm = None
        if self.clip_value is not None:
            for i, (_, o, n, tri) in enumerate(segs):
                _lib.check(L.l2hmc_grad_sumsq(gp + 4 * o, n, tri[0], tri[1], self._gnorm.data_ptr(), int(i > 0), s))
            if trainable_eps:
                _lib.check(L.l2hmc_grad_sumsq(gp + 4 * off, 1, 0, 0, self._gnorm.data_ptr(), 1, s))
            gnorm = self._gnorm.data_ptr()
        clip = self.clip_value if self.clip_value is not None else 0.
        for (wp, o, n, tri) in segs:
            _lib.check(L.l2hmc_adam_step(wp, gp + 4 * o, mp + 4 * o, vp + 4 * o, n, lr_t, self.beta1, self.beta2,
                                         self.epsilon, gnorm, clip, tri[0], tri[1], s))
        if trainable_eps:
            _lib.check(L.l2hmc_adam_step(self._eps_dev.data_ptr(), gp + 4 * off, mp + 4 * off, vp + 4 * off, 1, lr_t,
                                         self.beta1, self.beta2, self.epsilon, gnorm, clip, 0, 0, s))
            dyn.eps = self._eps_dev.detach().cpu().reshape(())     # the plan carries eps by value
        for net in self._nets:
            net.refresh_packed()
        self.global_step += 1

    def train_step(self, x, beta, **kw):
        """One evaluation of the reference's train_op: (loss, x_out, px, x_dq)."""
        out = self.calc_loss_and_grads(x, beta, **kw)
        self.apply_gradients()
        return out

    def update_beta(self, step, beta_init=2., beta_final=4., train_steps=10000):
        """gauge_model.py:1039-1046: linear annealing of 1/beta."""
        temp = (1. / beta_init - 1. / beta_final) * (1. - step / float(train_steps)) + 1. / beta_final
        return 1. / temp

    def train(self, train_steps, samples_init=None, beta_init=2., beta_final=4., initial_step=0, print_steps=0,
              log=print):
        """gauge_model.py:1119-1300 without the file / TensorBoard side effects: per step anneal beta, run the
        train op on the current samples, wrap the new samples to [0, 2 pi) (device side), record loss, accept
        probability, step size, learning rate and the observables of the step's INPUT samples (:256-266).
        Returns {'loss', 'accept_prob', 'eps', 'lr', 'beta', 'actions', 'plaqs', 'charges', 'charge_diff',
        'samples'} with per-step NumPy histories; the chain state never leaves the device inside the loop."""
        import numpy as np
        dyn = self.dynamics
        T, X = dyn.lattice.time_size, dyn.lattice.space_size
        if samples_init is None:
            samples_init = np.asarray(dyn.lattice.samples, dtype=np.float32).reshape(dyn.batch_size, dyn.x_dim)
        x = dyn._x(samples_init).clone()
        hist = {k: [] for k in ("loss", "accept_prob", "eps", "lr", "beta", "actions", "plaqs", "charges",
                                "charge_diff")}
        for step in range(initial_step, train_steps):
            beta = self.update_beta(step, beta_init, beta_final, train_steps)
            lr = self.learning_rate()
            obs = u1_observables(x, T, X)
            loss, x_out, px, x_dq = self.train_step(x, beta)
            _lib.check(_lib.lib().l2hmc_wrap_angle(x_out.data_ptr(), x_out.numel(), x.data_ptr(), _lib.stream_ptr(self.dynamics._device)))
            for k, v in (("loss", loss), ("accept_prob", px.mean()), ("actions", obs["action"].mean()),
                         ("plaqs", obs["avg_plaq"].mean()), ("charges", obs["top_charge"]),
                         ("charge_diff", x_dq.sum() / float(x_dq.numel()))):
                hist[k].append(v)
            hist["eps"].append(float(dyn.eps))
            hist["lr"].append(lr)
            hist["beta"].append(beta)
            if print_steps and step % print_steps == 0:
                log(f"{step:>5g}/{train_steps:<6g} loss {float(loss):^9.4g} acc {float(px.mean()):^9.4g} "
                    f"eps {float(dyn.eps):^9.4g} beta {beta:^9.4g} plaq {float(obs['avg_plaq'].mean()):^9.4g} "
                    f"lr {lr:^9.4g}")
        out = {k: (torch.stack(v).cpu().numpy() if v and isinstance(v[0], torch.Tensor) else np.asarray(v))
               for k, v in hist.items()}
        out["samples"] = x
        return out

    # ---- resumable state (gauge_model.py:519-556 `_current_state` + save_weights; .npz, never a pickle) ----
    def save_state(self, path, samples=None, beta=None):
        """Everything a resumed run needs: weights and Adam moments in the flat layout, step size, counters,
        masks, and optionally the chain state and beta of `_current_state`."""
        import numpy as np
        d = {"xnet": self._nets[0].flat_params()[0], "vnet": self._nets[1].flat_params()[0], "adam_m": self._m,
             "adam_v": self._v, "eps": self._eps_dev, "masks": self.dynamics.mask}
        out = {k: v.detach().cpu().numpy() for k, v in d.items()}
        out.update(global_step=np.int64(self.global_step), step=np.int64(self.global_step),
                   adam_t=np.int64(getattr(self, "_adam_t", 0)),
                   lr=np.float64(self.learning_rate()), draws=np.int64(self.dynamics._draws))
        if samples is not None:
            out["samples"] = samples.detach().cpu().numpy() if isinstance(samples, torch.Tensor) else np.asarray(samples)
        if beta is not None:
            out["beta"] = np.float64(beta)
        np.savez(path, **out)

    def load_state(self, path):
        """Inverse of save_state; returns {'samples', 'beta'} when they were stored."""
        import numpy as np
        dyn = self.dynamics
        with np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False) as f:
            for net, key in zip(self._nets, ("xnet", "vnet")):
                flat = net.flat_params()[0]
                if f[key].shape != tuple(flat.shape):
                    raise ValueError(f"{key}: stored {f[key].shape}, expected {tuple(flat.shape)}")
                flat.copy_(torch.from_numpy(f[key]))
                net.pack()
                net.refresh_packed()
            self._m.copy_(torch.from_numpy(f["adam_m"]))
            self._v.copy_(torch.from_numpy(f["adam_v"]))
            self._eps_dev.copy_(torch.from_numpy(f["eps"]))
            dyn.eps = self._eps_dev.detach().cpu().reshape(())
            dyn.set_masks(f["masks"])
            self.global_step, self._adam_t = int(f["global_step"]), int(f["adam_t"])
            dyn._draws = int(f["draws"])
            extra = {k: f[k] for k in ("samples", "beta") if k in f.files}
        self.sync_weights()
        return extra

    def current_state(self, samples=None, beta=None):
        """The reference's `_current_state` dict (gauge_model.py:370-376, :1190-1194) with its own keys --
        {'samples', 'eps', 'step', 'beta', 'lr'} -- from this trainer; `save_state` writes the same keys (plus the
        optimiser state the reference leaves to the TF checkpoint) into its .npz."""
        import numpy as np
        if isinstance(samples, torch.Tensor):
            samples = samples.detach().cpu().numpy()
        return {'samples': samples, 'eps': float(self.dynamics.eps), 'step': int(self.global_step),
                'beta': None if beta is None else float(beta), 'lr': float(self.learning_rate() / self.world)}

    @staticmethod
    def train_data_dict(out, initial_step=0):
        """`train()` histories -> the reference's `train_data_dict` (gauge_model.py:362-369, :1196-1205):
        {'loss' | 'actions' | 'plaqs' | 'charges' | 'charge_diff' | 'accept_prob': {(step, beta): value}}."""
        keys = [(initial_step + i, float(b)) for i, b in enumerate(out['beta'])]
        return {name: {k: out[name][i] for i, k in enumerate(keys)}
                for name in ('loss', 'actions', 'plaqs', 'charges', 'charge_diff', 'accept_prob')}

    def sync_weights(self):
        """Bring the reference-layout layer tensors (state_dict / save_weights) up to date."""
        for net in self._nets:
            net.sync_reference_layout()
