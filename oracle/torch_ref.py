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

    def __init__(self, T, X, num_steps, eps, masks, xnet, vnet, arch='generic'):
        self.arch = arch
        self.T, self.X, self.N = T, X, num_steps
        self.eps = torch.tensor(float(eps), dtype=torch.float64, requires_grad=True)
        self.mask = torch.tensor(np.asarray(masks), dtype=torch.float64)
        self.xnet = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in xnet.items()}
        self.vnet = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in vnet.items()}

    def _net(self, p, inputs):
        return generic_net(p, inputs) if self.arch == 'generic' else conv3d_net(p, inputs, self.T, self.X)

    def parameters(self):
        return [self.eps] + [self.xnet[k] for k in sorted(self.xnet)] + [self.vnet[k] for k in sorted(self.vnet)]

    def _time(self, i, B):
        arg = TWO_PI * i / self.N
        return torch.tensor([[np.cos(arg), np.sin(arg)]], dtype=torch.float64).repeat(B, 1)

    def _upd_v(self, x, v, beta, t, bwd):
        g = beta * grad_action(x, self.T, self.X)
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
        ld = torch.zeros(x.shape[0], dtype=torch.float64)
        for step in range(self.N):
            x, v, j = self.leapfrog(x, v, beta, step, bwd)
            ld = ld + j
        h0 = beta * action(x0, self.T, self.X) + 0.5 * (v0 ** 2).sum(1)
        h1 = beta * action(x, self.T, self.X) + 0.5 * (v ** 2).sum(1)
        p = torch.exp(torch.minimum(h0 - h1 + ld, torch.zeros((), dtype=torch.float64)))
        return x, v, torch.where(torch.isfinite(p), p, torch.zeros_like(p)), ld

    def apply_transition(self, x, beta, v0f, v0b, coin, u):
        xf, vf, pf, _ = self.trajectory(x, v0f, beta, False)
        xb, vb, pb, _ = self.trajectory(x, v0b, beta, True)
        fm = (coin > 0.5).double()
        bm = 1. - fm
        xp = fm[:, None] * xf + bm[:, None] * xb
        vp = fm[:, None] * vf + bm[:, None] * vb
        p = fm * pf + bm * pb
        am = (p > u).double()
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
