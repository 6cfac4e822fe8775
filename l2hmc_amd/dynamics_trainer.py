"""Training step of the generic L2HMC sampler on the toy targets (second half):
  l2hmc/mog_model.py:324-355  _create_loss   (squared jump distance of the x and z chains)
  l2hmc/mog_model.py:357-363  _create_optimizer (AdamOptimizer.minimize)
  l2hmc/mog_model.py:183-192  exponential_decay learning rate
Forward, loss and the whole reverse pass run in ONE library call (l2hmc_small_train_step); Adam is the same
flat-buffer kernel the lattice trainer uses.  Chains run in the direction `propose` picks for them
(sampler.py:35-41 multiplies the other direction by an exact 0)."""
import ctypes as C

import torch

from . import _lib
from .dist import active as _active_dist


class DynamicsTrainer:
    def __init__(self, dynamics, lr_init=1e-2, lr_decay_steps=2500, lr_decay_rate=0.96, scale=0.1, dist=None,
                 beta1=0.9, beta2=0.999, epsilon=1e-8):
        if dynamics.hmc:
            raise ValueError("hmc=True dynamics have no trainable networks")
        self.dynamics = dyn = dynamics
        self.scale = float(scale)
        self.lr_init, self.lr_decay_steps, self.lr_decay_rate = float(lr_init), int(lr_decay_steps), float(lr_decay_rate)
        self.beta1, self.beta2, self.epsilon = float(beta1), float(beta2), float(epsilon)
        self.dist = _active_dist(dist)
        self.world = self.dist.get_world_size() if self.dist is not None else 1
        self.global_step, self._adam_t = 0, 0
        dev = dyn._device
        self._nets = (dyn.XNet, dyn.VNet)
        flats = [n.flat_params() for n in self._nets]
        self._sizes = [f[0].numel() for f in flats]
        n_all = sum(self._sizes) + 1
        self.grads = torch.zeros(n_all, dtype=torch.float32, device=dev)      # [xnet | vnet | alpha]
        self._m, self._v = torch.zeros_like(self.grads), torch.zeros_like(self.grads)
        self._alpha_dev = dyn.alpha.detach().to(dev, torch.float32).reshape(1).clone()
        self._ws = _lib.Workspace()

    def grad_views(self):
        out, off = {}, 0
        for name, net in zip(("xnet", "vnet"), self._nets):
            flat, views, offsets = net.flat_params()
            out[name] = {k: self.grads[off + a:off + b].view(views[k].shape) for k, (a, b) in offsets.items()}
            off += flat.numel()
        out["alpha"] = self.grads[off:off + 1]
        return out

    def learning_rate(self):
        return self.lr_init * self.lr_decay_rate ** (self.global_step // self.lr_decay_steps)

    def calc_loss_and_grads(self, x, z=None, draws_x=None, draws_z=None):
        """-> (loss, x_out, px).  draws_*: optional (init_v_forward, init_v_backward, dir_bits (1 = forward), u)."""
        dyn = self.dynamics
        dev = dyn._device
        x = _lib.as_dev(x, dev).reshape(-1, dyn.x_dim)
        B, D = x.shape
        z = dyn._normal((B, D)) if z is None else _lib.as_dev(z, dev).reshape(B, D)

        def uniform(n):
            out = torch.empty(n, dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().l2hmc_fill_uniform(out.data_ptr(), n, dyn._seed, dyn._draws, _lib.stream_ptr(self.dynamics._device)))
            dyn._draws += 1
            return out

        def draw(d):
            if d is None:
                return dyn._normal((B, D)), dyn._normal((B, D)), (uniform(B) >= 0.5).to(torch.float32), uniform(B)
            return tuple(_lib.as_dev(a, dev) for a in d)
        vf_x, vb_x, bits_x, u_x = draw(draws_x)
        vf_z, vb_z, bits_z, _ = draw(draws_z)
        fwd = torch.cat([bits_x, bits_z]) > 0.5
        x0 = torch.cat([x, z]).contiguous()
        v0 = torch.where(fwd[:, None], torch.cat([vf_x, vf_z]), torch.cat([vb_x, vb_z])).contiguous()
        dirs = (~fwd).to(torch.int32).contiguous()
        R = 2 * B
        xN, vN = torch.empty_like(x0), torch.empty_like(x0)
        p, terms = (torch.empty(R, dtype=torch.float32, device=dev) for _ in range(2))
        plan, L = dyn._plan(), _lib.lib()
        ws, nb = self._ws.get(L.l2hmc_small_train_ws_bytes(C.byref(plan), R), dev)
        _lib.check(L.l2hmc_small_train_step(C.byref(plan), x0.data_ptr(), v0.data_ptr(), dirs.data_ptr(), R, self.scale,
                                            1.0 / (B * self.world), xN.data_ptr(), vN.data_ptr(), p.data_ptr(),
                                            terms.data_ptr(), self.grads.data_ptr(), ws, nb, _lib.stream_ptr(self.dynamics._device)))
        self.grads[-1] *= float(dyn.eps)              # d/d alpha = eps * d/d eps  (utils/dynamics.py:51-60)
        buf = torch.stack([terms.sum(dtype=torch.float32), torch.full((), float(B), dtype=torch.float32, device=dev)])
        if self.dist is not None:
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
            self.dist.all_reduce(self.grads, op=self.dist.ReduceOp.SUM)
        loss = buf[0] / buf[1]
        px, Lx = p[:B], xN[:B]
        x_out = torch.where(((px - u_x) >= 0)[:, None], Lx, x)       # sampler.py:57-59
        self.last_terms, self.last_proposals, self.last_p = terms, xN, p
        return loss, x_out, px

    def apply_gradients(self):
        dyn, L, s = self.dynamics, _lib.lib(), _lib.stream_ptr(self.dynamics._device)
        lr = self.learning_rate()
        self._adam_t += 1
        t = self._adam_t
        lr_t = lr * (1. - self.beta2 ** t) ** 0.5 / (1. - self.beta1 ** t)
        gp, mp, vp = self.grads.data_ptr(), self._m.data_ptr(), self._v.data_ptr()
        off = 0
        for net in self._nets:
            flat, _, offsets = net.flat_params()
            tri = offsets["b1"]
            _lib.check(L.l2hmc_adam_step(flat.data_ptr(), gp + 4 * off, mp + 4 * off, vp + 4 * off, flat.numel(), lr_t,
                                         self.beta1, self.beta2, self.epsilon, None, 0., tri[0], tri[1], s))
            off += flat.numel()
        if dyn.eps_trainable:
            _lib.check(L.l2hmc_adam_step(self._alpha_dev.data_ptr(), gp + 4 * off, mp + 4 * off, vp + 4 * off, 1, lr_t,
                                         self.beta1, self.beta2, self.epsilon, None, 0., 0, 0, s))
            dyn.alpha = self._alpha_dev.detach().cpu().reshape(())
        for net in self._nets:
            net.refresh_packed()
        self.global_step += 1

    def train_step(self, x, **kw):
        out = self.calc_loss_and_grads(x, **kw)
        self.apply_gradients()
        return out

    def sync_weights(self):
        for net in self._nets:
            net.sync_reference_layout()
