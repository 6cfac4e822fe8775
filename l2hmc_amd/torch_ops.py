"""`torch.ops.l2hmc.*`: the stateless operators of the C ABI registered with PyTorch's dispatcher (SURVEY.md 8b names
`extern "C"` / `TORCH_LIBRARY(l2hmc)` as the boundary; the C ABI of include/l2hmc_hip.h is the boundary, this file makes
the same entry points visible to callers that compose torch operators).  Every op takes contiguous fp32 CUDA (ROCm)
tensors, launches on torch's current stream, allocates only its outputs and has NO CPU implementation: called with a CPU
tensor it raises like every other entry of this package.  Networks travel as the flat tuple of their packed device
buffers (`_DenseSTQ.flat_tensors()` order: w1_t, wt, b1, wh_t, bh, whd_t, bhd, coeff_s, coeff_q).

    import l2hmc_amd.torch_ops            # registers the library
    action, force, plaq, charge = torch.ops.l2hmc.u1_action_force(x, T, X, beta)
    S, T_, Q = torch.ops.l2hmc.stq_dense(a, b, weights, q_tanh, t_cos, t_sin)
    v1, logdet = torch.ops.l2hmc.lf_update_v(v, grad, S, T_, Q, eps, direction)
    x1, logdet = torch.ops.l2hmc.lf_update_x(x, v, keep, S, T_, Q, eps, direction)
    p = torch.ops.l2hmc.accept_prob(h_old, h_new, sumlogdet)
    x_prop, v_prop, p, x_out = torch.ops.l2hmc.mix_accept(x, xf, vf, pf, xb, vb, pb, coin, u, strict)
    k = torch.ops.l2hmc.kinetic_energy(v)
    y = torch.ops.l2hmc.wrap_angle(x)
"""
import ctypes as C
from typing import List, Tuple

import torch

from . import _lib

_WEIGHT_FIELDS = ("w1_t", "wt", "b1", "wh_t", "bh", "whd_t", "bhd", "coeff_s", "coeff_q")


def _f32(t, name):
    return _lib.dev_ptr(t, name=name)


def _net_struct(weights, q_tanh):
    if len(weights) != len(_WEIGHT_FIELDS):
        raise ValueError(f"weights: expected {len(_WEIGHT_FIELDS)} tensors {_WEIGHT_FIELDS}, got {len(weights)}")
    w1_t, wt, _, wh_t, _, whd_t = weights[:6]
    H, K = w1_t.shape
    D = whd_t.shape[1]
    if wh_t.shape != (H, H) or whd_t.shape != (3, D, H) or tuple(wt.shape) != (2, H) or K % 2:
        raise ValueError("weights: shapes do not form a packed S/T/Q network (include/l2hmc_hip.h: l2hmc_dense_net)")
    st = _lib.DenseNet(D=D, H=H, Ka=K // 2, Kb=K // 2, q_tanh=int(q_tanh), reserved=0, packed=None)
    for f, t in zip(_WEIGHT_FIELDS, weights):
        setattr(st, f, _f32(t, f))
    return st


@torch.library.custom_op("l2hmc::u1_action_force", mutates_args=())
def u1_action_force(x: torch.Tensor, time_size: int, space_size: int, beta: float) -> Tuple[torch.Tensor, torch.Tensor,
                                                                                             torch.Tensor, torch.Tensor]:
    """lattice.py:285-362, gauge_dynamics.py:698-709: (action, beta * dS/dx, average plaquette, topological charge)."""
    D = 2 * time_size * space_size
    x = x.reshape(-1, D)
    rows = x.shape[0]
    action, plaq, charge = (torch.empty(rows, dtype=torch.float32, device=x.device) for _ in range(3))
    force = torch.empty_like(x)
    _lib.check(_lib.lib().l2hmc_u1_action_force(_f32(x, "x"), rows, time_size, space_size, float(beta), action.data_ptr(),
                                                force.data_ptr(), plaq.data_ptr(), charge.data_ptr(),
                                                _lib.stream_ptr(x.device)))
    return action, force, plaq, charge


@torch.library.custom_op("l2hmc::stq_dense", mutates_args=())
def stq_dense(a: torch.Tensor, b: torch.Tensor, weights: List[torch.Tensor], q_tanh: int, t_cos: float,
              t_sin: float) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """generic_net.py:129-146 / utils/network.py:89-114 (q_tanh = 1): (S, T, Q) = net([a, b, t])."""
    st = _net_struct(weights, q_tanh)
    rows = a.shape[0]
    S, T, Q = (torch.empty(rows, st.D, dtype=torch.float32, device=a.device) for _ in range(3))
    L = _lib.lib()
    nb = L.l2hmc_stq_ws_bytes(rows, st.H)
    ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=a.device)
    _lib.check(L.l2hmc_stq_dense(C.byref(st), _f32(a, "a"), _f32(b, "b"), None, float(t_cos), float(t_sin), rows,
                                 S.data_ptr(), T.data_ptr(), Q.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr(a.device)))
    return S, T, Q


@torch.library.custom_op("l2hmc::lf_update_v", mutates_args=())
def lf_update_v(v: torch.Tensor, grad: torch.Tensor, S: torch.Tensor, T: torch.Tensor, Q: torch.Tensor, eps: float,
                direction: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """gauge_dynamics.py:486-508 (direction 0), :537-561 (1): (v', per-row log-det)."""
    out, ld = torch.empty_like(v), torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
    _lib.check(_lib.lib().l2hmc_lf_update_v(_f32(v, "v"), _f32(grad, "grad"), _f32(S, "S"), _f32(T, "T"), _f32(Q, "Q"),
                                            float(eps), int(direction), v.shape[0], v.shape[1], out.data_ptr(),
                                            ld.data_ptr(), _lib.stream_ptr(v.device)))
    return out, ld


@torch.library.custom_op("l2hmc::lf_update_x", mutates_args=())
def lf_update_x(x: torch.Tensor, v: torch.Tensor, keep: torch.Tensor, S: torch.Tensor, T: torch.Tensor, Q: torch.Tensor,
                eps: float, direction: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """gauge_dynamics.py:511-534 (direction 0), :565-590 (1); keep [D]: 1 = coordinate kept."""
    out, ld = torch.empty_like(x), torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().l2hmc_lf_update_x(_f32(x, "x"), _f32(v, "v"), _f32(keep, "keep"), _f32(S, "S"), _f32(T, "T"),
                                            _f32(Q, "Q"), float(eps), int(direction), x.shape[0], x.shape[1],
                                            out.data_ptr(), ld.data_ptr(), _lib.stream_ptr(x.device)))
    return out, ld


@torch.library.custom_op("l2hmc::accept_prob", mutates_args=())
def accept_prob(h_old: torch.Tensor, h_new: torch.Tensor, sumlogdet: torch.Tensor) -> torch.Tensor:
    """gauge_dynamics.py:592-609: exp(min(h_old - h_new + sumlogdet, 0)), non-finite -> 0."""
    p = torch.empty_like(h_old)
    _lib.check(_lib.lib().l2hmc_accept_prob(_f32(h_old, "h_old"), _f32(h_new, "h_new"), _f32(sumlogdet, "sumlogdet"),
                                            p.numel(), p.data_ptr(), _lib.stream_ptr(p.device)))
    return p


@torch.library.custom_op("l2hmc::mix_accept", mutates_args=())
def mix_accept(x: torch.Tensor, xf: torch.Tensor, vf: torch.Tensor, pf: torch.Tensor, xb: torch.Tensor, vb: torch.Tensor,
               pb: torch.Tensor, coin: torch.Tensor, u: torch.Tensor, strict: int) -> Tuple[torch.Tensor, torch.Tensor,
                                                                                              torch.Tensor, torch.Tensor]:
    """gauge_dynamics.py:221-257 (strict = 1) / utils/sampler.py:33-59 (0): (x_prop, v_prop, p, x_out)."""
    x_prop, v_prop, x_out = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    p = torch.empty_like(pf)
    _lib.check(_lib.lib().l2hmc_mix_accept(
        _f32(x, "x"), _f32(xf, "xf"), _f32(vf, "vf"), _f32(pf, "pf"), _f32(xb, "xb"), _f32(vb, "vb"), _f32(pb, "pb"),
        _f32(coin, "coin"), _f32(u, "u"), int(strict), x.shape[0], x.shape[1], x_prop.data_ptr(), v_prop.data_ptr(),
        p.data_ptr(), x_out.data_ptr(), _lib.stream_ptr(x.device)))
    return x_prop, v_prop, p, x_out


@torch.library.custom_op("l2hmc::kinetic_energy", mutates_args=())
def kinetic_energy(v: torch.Tensor) -> torch.Tensor:
    """gauge_dynamics.py:683-689: 0.5 * sum_d v^2 per row."""
    out = torch.empty(v.shape[0], dtype=torch.float32, device=v.device)
    _lib.check(_lib.lib().l2hmc_kinetic_energy(_f32(v, "v"), v.shape[0], v.shape[1], out.data_ptr(),
                                               _lib.stream_ptr(v.device)))
    return out


@torch.library.custom_op("l2hmc::wrap_angle", mutates_args=())
def wrap_angle(x: torch.Tensor) -> torch.Tensor:
    """gauge_model.py:1180, :1388: x mod 2 pi in [0, 2 pi)."""
    out = torch.empty_like(x)
    _lib.check(_lib.lib().l2hmc_wrap_angle(_f32(x, "x"), x.numel(), out.data_ptr(), _lib.stream_ptr(x.device)))
    return out


OPS = ("u1_action_force", "stq_dense", "lf_update_v", "lf_update_x", "accept_prob", "mix_accept", "kinetic_energy",
       "wrap_angle")
