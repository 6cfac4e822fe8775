"""Oracle (test infrastructure): differentiable torch-CPU restatement of the lattice model's training
graph -- the checker for gradients.

Same arithmetic as oracle/gauge_dynamics.py and oracle/loss.py (which restate
l2hmc/dynamics/gauge_dynamics.py:195-313,412-609 and l2hmc/gauge_model.py:728-832), written with torch
float64 ops so that torch.autograd supplies what the reference gets from tf.gradients
(gauge_model.py:825).  The force is the closed form of grad_potential (not a nested autograd call), which
is what tf.gradients of the roll/cos action evaluates to; autograd then differentiates through it like TF
differentiates through its own gradient graph.
"""
import numpy as np
import torch

TWO_PI = 2 * np.pi


def plaq(x, T, X):
    s = x.reshape(x.shape[0], T, X, 2)
    return s[..., 0] - s[..., 1] - torch.roll(s[..., 0], -1, 2) + torch.roll(s[..., 1], -1, 1)


def action(x, T, X):
    return torch.sum(1. - torch.cos(plaq(x, T, X)), dim=(1, 2))


def grad_action(x, T, X):
    sp = torch.sin(plaq(x, T, X))
    g0 = sp - torch.roll(sp, 1, 2)
    g1 = -sp + torch.roll(sp, 1, 1)
    return torch.stack([g0, g1], dim=-1).reshape(x.shape[0], -1)


def generic_net(p, inputs):
    v, x, t = inputs
    h = (v @ p['v_layer/W'] + p['v_layer/b']) + (x @ p['x_layer/W'] + p['x_layer/b']) \
        + (t @ p['t_layer/W'] + p['t_layer/b'])
    h = torch.relu(h)
    h = torch.relu(h @ p['h_layer/W'] + p['h_layer/b'])
    S = torch.tanh(h @ p['scale_layer/W'] + p['scale_layer/b']) * torch.exp(p['coeff_scale'])
    Tr = h @ p['translation_layer/W'] + p['translation_layer/b']
    Q = (h @ p['transformation_layer/W'] + p['transformation_layer/b']) * torch.exp(p['coeff_transformation'])
    return S, Tr, Q


def conv3d_front(p, a, which, T, X):
    """network/conv_net.py:251-262 for one input (channels_last): reshape_5D -> Conv3D(F,(3,3,2),same,relu) ->
    MaxPool3D(2,s2,same) -> Conv3D(2F,(2,2,2),same,relu) -> MaxPool3D(same) -> flatten.  Keras conventions as
    in oracle/nets.py:134-170 ('same' pads floor((k-1)/2) before, the rest after; pooling clips at the border)."""
    import torch.nn.functional as Fn
    B = a.shape[0]
    h = a.reshape(B, 1, T, X, 2)                                    # torch layout [B, C, T, X, depth]
    w1 = p[f'conv_{which}1/W'].permute(4, 3, 0, 1, 2)               # Keras [kh,kw,kd,Cin,Cout] -> [Cout,Cin,kh,kw,kd]
    h = torch.relu(Fn.conv3d(Fn.pad(h, (0, 1, 1, 1, 1, 1)), w1, p[f'conv_{which}1/b']))
    h = Fn.max_pool3d(h, (2, 2, 2))                                 # depth 2 -> 1
    w2 = p[f'conv_{which}2/W'].permute(4, 3, 0, 1, 2)
    h = torch.relu(Fn.conv3d(Fn.pad(h, (0, 1, 0, 1, 0, 1)), w2, p[f'conv_{which}2/b']))
    h = Fn.max_pool3d(h, (2, 2, 1))                                 # depth 1: the 'same' window holds one cell
    return h.permute(0, 2, 3, 4, 1).reshape(B, -1)                  # flatten over (h, w, depth, channel)


def conv3d_net(p, inputs, T, X):
    v, x, t = inputs
    return generic_net(p, [conv3d_front(p, v, 'v', T, X), conv3d_front(p, x, 'x', T, X), t])


class TorchGaugeModel:
    """Weights and eps are leaf tensors with requires_grad=True."""

    def __init__(self, T, X, num_steps, eps, masks, xnet, vnet, arch='generic', dtype=torch.float64):
        # dtype=torch.float32 evaluates the same graph in the reference's precision: the yardstick for how far an
        # fp32 implementation may sit from the float64 answer on chaotic trajectories
        self.arch, self.dtype = arch, dtype
        self.T, self.X, self.N = T, X, num_steps
        self.eps = torch.tensor(float(eps), dtype=dtype, requires_grad=True)
        self.mask = torch.tensor(np.asarray(masks), dtype=dtype)
        self.xnet = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in xnet.items()}
        self.vnet = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in vnet.items()}

    def _force(self, x, beta):
        return beta * grad_action(x, self.T, self.X)

    def _energy(self, x, beta):
        return beta * action(x, self.T, self.X)

    def _net(self, p, inputs):
        return generic_net(p, inputs) if self.arch == 'generic' else conv3d_net(p, inputs, self.T, self.X)

    def parameters(self):
        return [self.eps] + [self.xnet[k] for k in sorted(self.xnet)] + [self.vnet[k] for k in sorted(self.vnet)]

    def _time(self, i, B):
        arg = TWO_PI * i / self.N
        return torch.tensor([[np.cos(arg), np.sin(arg)]], dtype=self.dtype).repeat(B, 1)

    def _upd_v(self, x, v, beta, t, bwd):
        g = self._force(x, beta)
        S, Tr, Q = self._net(self.vnet, [x, g, t])
        eps = self.eps
        if not bwd:
            s = S * (0.5 * eps)
            return v * torch.exp(s) - 0.5 * eps * (torch.exp(Q * eps) * g - Tr), s.sum(1)
        s = S * (-0.5 * eps)
        return torch.exp(s) * (v + 0.5 * eps * (torch.exp(Q * eps) * g - Tr)), s.sum(1)

    def _upd_x(self, x, v, t, m, mi, bwd):
        S, Tr, Q = self._net(self.xnet, [v, m * x, t])
        eps = self.eps
        if not bwd:
            s = S * eps
            tmp = x * torch.exp(s) + eps * (torch.exp(Q * eps) * v + Tr)
        else:
            s = S * (-eps)
            tmp = torch.exp(s) * (x - eps * (torch.exp(Q * eps) * v + Tr))
        return m * x + mi * tmp, (mi * s).sum(1)

    def leapfrog(self, x, v, beta, step, bwd):
        i = self.N - step - 1 if bwd else step
        t = self._time(i, x.shape[0])
        m = self.mask[i]
        mi = 1. - m
        v, l1 = self._upd_v(x, v, beta, t, bwd)
        if not bwd:
            x, l2 = self._upd_x(x, v, t, m, mi, bwd)
            x, l3 = self._upd_x(x, v, t, mi, m, bwd)
        else:
            x, l2 = self._upd_x(x, v, t, mi, m, bwd)
            x, l3 = self._upd_x(x, v, t, m, mi, bwd)
        v, l4 = self._upd_v(x, v, beta, t, bwd)
        return x, v, l1 + l2 + l3 + l4

    def trajectory(self, x0, v0, beta, bwd):
        x, v = x0, v0
        ld = torch.zeros(x.shape[0], dtype=self.dtype)
        for step in range(self.N):
            x, v, j = self.leapfrog(x, v, beta, step, bwd)
            ld = ld + j
        h0 = self._energy(x0, beta) + 0.5 * (v0 ** 2).sum(1)
        h1 = self._energy(x, beta) + 0.5 * (v ** 2).sum(1)
        p = torch.exp(torch.minimum(h0 - h1 + ld, torch.zeros((), dtype=self.dtype)))
        return x, v, torch.where(torch.isfinite(p), p, torch.zeros_like(p)), ld

    def apply_transition(self, x, beta, v0f, v0b, coin, u):
        xf, vf, pf, _ = self.trajectory(x, v0f, beta, False)
        xb, vb, pb, _ = self.trajectory(x, v0b, beta, True)
        fm = (coin > 0.5).to(self.dtype)
        bm = 1. - fm
        xp = fm[:, None] * xf + bm[:, None] * xb
        vp = fm[:, None] * vf + bm[:, None] * vb
        p = fm * pf + bm * pb
        am = (p > u).to(self.dtype)
        return xp, vp, p, am[:, None] * xp + (1. - am)[:, None] * x

    def loss(self, x, z, beta, draws_x, draws_z, metric='cos_diff', loss_scale=1., aux_weight=1., std_weight=1.,
             charge_weight=1.):
        """gauge_model.py:728-797; returns (loss, per-chain terms)."""
        T, X = self.T, self.X
        x_, _, px, _ = self.apply_transition(x, beta, *draws_x)
        _, _, pz, _ = self.apply_transition(z, beta, *draws_z)
        eps = 1e-3
        m = {'l1': lambda a, b: torch.abs(a - b), 'l2': lambda a, b: (a - b) ** 2,
             'cos': lambda a, b: torch.abs(torch.cos(a) - torch.cos(b)),
             'cos2': lambda a, b: (torch.cos(a) - torch.cos(b)) ** 2,
             'cos_diff': lambda a, b: 1. - torch.cos(a - b)}[metric]

        def q_fft(a):
            pq = plaq(a, T, X)
            y = torch.zeros_like(pq)
            for n in range(1, 5):
                y = y + (-2. / n) * ((-1.) ** n) * torch.sin(n * pq)
            return y.sum(dim=(1, 2)) / TWO_PI

        x_std = m(x, x_).sum(1) * px + eps
        z_std = aux_weight * (m(z, x_).sum(1) * pz + eps)
        std_loss = std_weight * (loss_scale * (1. / x_std + 1. / z_std) - (x_std + z_std) / loss_scale)
        xq = px * torch.abs(q_fft(x) - q_fft(x_)) + eps
        zq = aux_weight * (pz * torch.abs(q_fft(z) - q_fft(x_)) + eps)
        terms = std_loss + charge_weight * (xq + zq)
        return terms.mean(), terms


# ---------------------------------------------------------------------------------------------
# generic Dynamics on the toy targets (utils/dynamics.py:120-319, utils/network.py:89-114,
# utils/distributions.py:151-158) and the loss of mog_model.py:324-355
# ---------------------------------------------------------------------------------------------
def mlp_net(p, inputs):
    a, b, t = inputs
    h = (a @ p['embed_1/W'] + p['embed_1/b']) + (b @ p['embed_2/W'] + p['embed_2/b']) \
        + (t @ p['embed_3/W'] + p['embed_3/b'])
    h = torch.relu(h)
    h = torch.relu(h @ p['linear_1/W'] + p['linear_1/b'])
    S = torch.exp(p['scale_s']) * torch.tanh(h @ p['linear_s/W'] + p['linear_s/b'])
    Tr = h @ p['linear_t/W'] + p['linear_t/b']
    Fq = torch.exp(p['scale_f']) * torch.tanh(h @ p['linear_f/W'] + p['linear_f/b'])
    return S, Tr, Fq


class TorchDynamicsModel(TorchGaugeModel):
    """Same sub-update algebra as the lattice class (the two reference files write identical formulas);
    differences: MLP with tanh on both S and F, eps = exp(alpha) with alpha the trainable leaf, mixture
    energy with closed-form gradient (what tf.gradients of :151-158 evaluates to), temperature."""

    def __init__(self, target, num_steps, eps, masks, xnet, vnet, temperature=1.0):
        self.arch, self.N, self.dtype = 'mlp', num_steps, torch.float64
        self.alpha = torch.tensor(float(np.log(eps)), dtype=torch.float64, requires_grad=True)
        self.mask = torch.tensor(np.asarray(masks), dtype=torch.float64)
        self.xnet = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in xnet.items()}
        self.vnet = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in vnet.items()}
        self.temperature = float(temperature)
        if hasattr(target, 'i_sigmas'):             # oracle.dynamics.GMM
            self.mus = [torch.tensor(m.astype('float32').astype('float64')) for m in target.mus]
            self.precs = [torch.tensor(s.astype('float64')) for s in target.i_sigmas]
            self.logc = [float(np.log(c)) for c in target.constants]
            self.gaussian = False
        else:                                       # oracle.dynamics.Gaussian
            self.mus = [torch.tensor(target.mu.astype('float32').astype('float64'))]
            self.precs = [torch.tensor(target.i_sigma.astype('float32').astype('float64'))]
            self.logc = [0.0]
            self.gaussian = True

    @property
    def eps(self):
        return torch.exp(self.alpha)

    def _net(self, p, inputs):
        return mlp_net(p, inputs)

    def _V(self, x):
        cols = []
        for mu, P, c in zip(self.mus, self.precs, self.logc):
            d = x - mu
            cols.append(-0.5 * torch.einsum('bi,ij,bj->b', d, P, d) + c)
        return torch.stack(cols, dim=1)

    def _energy(self, x, beta):
        V = self._V(x)
        e = -V[:, 0] if self.gaussian else -torch.logsumexp(V, dim=1)
        return e / self.temperature

    def _force(self, x, beta):
        w = torch.ones(x.shape[0], 1, dtype=torch.float64) if self.gaussian else torch.softmax(self._V(x), dim=1)
        g = torch.zeros_like(x)
        for i, (mu, P) in enumerate(zip(self.mus, self.precs)):
            g = g + w[:, i:i + 1] * (0.5 * (x - mu) @ (P + P.T))
        return g / self.temperature

    def propose(self, x, v0f, v0b, bits):
        xf, vf, pf, _ = self.trajectory(x, v0f, None, False)
        xb, vb, pb, _ = self.trajectory(x, v0b, None, True)
        m = bits.double()
        return m[:, None] * xf + (1 - m)[:, None] * xb, m * pf + (1 - m) * pb

    def mog_loss(self, x, z, draws_x, draws_z, scale):
        """mog_model.py:336-355."""
        Lx, px = self.propose(x, draws_x[0], draws_x[1], draws_x[2])
        Lz, pz = self.propose(z, draws_z[0], draws_z[1], draws_z[2])
        v1 = ((x - Lx) ** 2).sum(1) * px + 1e-4
        v2 = ((z - Lz) ** 2).sum(1) * pz + 1e-4
        loss = scale * ((1.0 / v1).mean() + (1.0 / v2).mean()) + (-v1.mean() - v2.mean()) / scale
        return loss, Lx, px, Lz, pz
