"""Sampler glue with the reference's signatures (l2hmc/utils/sampler.py:28-59).
Random draws (direction bits, MH uniforms, momenta) come from the library's
Philox stream unless injected through the keyword-only arguments."""
import ctypes as C

import torch

from . import _lib


def _uniform(dynamics, n):
    out = torch.empty(n, dtype=torch.float32, device=dynamics._device)
    _lib.check(_lib.lib().l2hmc_fill_uniform(out.data_ptr(), n, dynamics._seed, dynamics._draws,
                                             _lib.stream_ptr(out.device)))
    dynamics._draws += 1
    return out


def tf_accept(x, Lx, px, u=None, dynamics=None):
    """:57-59 -- accept iff px - u >= 0."""
    x, Lx, px = (_lib.as_dev(t) for t in (x, Lx, px))
    u = _lib.as_dev(u) if u is not None else _uniform(dynamics, px.numel())
    out = torch.empty_like(x)
    ones = torch.ones_like(px)
    # forward slot carries Lx with coin=1; strict=0 selects the sampler.py comparison
    _lib.check(_lib.lib().l2hmc_mix_accept(
        x.data_ptr(), Lx.data_ptr(), Lx.data_ptr(), px.data_ptr(), Lx.data_ptr(), Lx.data_ptr(), px.data_ptr(),
        ones.data_ptr(), u.data_ptr(), 0, x.shape[0], x.shape[1], None, None, None, out.data_ptr(),
        _lib.stream_ptr(x.device)))
    return out


def propose(x, dynamics, init_v=None, aux=None, do_mh_step=False, log_jac=False, *,
            init_v_backward=None, dir_bits=None, u=None):
    """:28-55 -> (Lx, Lv, px, outputs)."""
    x = _lib.as_dev(x, dynamics._device)
    if dynamics.hmc:
        Lx, Lv, px = dynamics.forward(x, init_v=init_v, aux=aux)
        return Lx, Lv, px, [tf_accept(x, Lx, px, u, dynamics)]
    B = x.shape[0]
    if aux is not None:
        raise NotImplementedError("aux inputs are only used by the out-of-scope VAE scripts")
    if (init_v is None and init_v_backward is None and dir_bits is None and u is None and not log_jac
            and not dynamics.layered):
        # every draw is the library's: ONE launch (direction bit, both momenta, both trajectories, mix, MH);
        # same Philox streams, same numbers as the piecewise path below
        x = x.reshape(-1, dynamics.x_dim).contiguous()
        Lx, px = torch.empty_like(x), torch.empty(B, dtype=torch.float32, device=x.device)
        out = torch.empty_like(x) if do_mh_step else None
        plan = dynamics._plan()
        _lib.check(_lib.lib().l2hmc_small_propose(
            C.byref(plan), x.data_ptr(), B, dynamics._seed, dynamics._draws, Lx.data_ptr(), None, px.data_ptr(),
            None if out is None else out.data_ptr(), _lib.stream_ptr(x.device)))
        dynamics._draws += 4 if do_mh_step else 3
        return Lx, None, px, ([out] if do_mh_step else [])       # Lv is None without init_v (:43-45, quirk Q6)
    if dir_bits is None:
        mask = (_uniform(dynamics, B) >= 0.5).to(torch.float32)     # randint{0,1}
    else:
        mask = _lib.as_dev(dir_bits, dynamics._device)
    vb = init_v_backward if init_v_backward is not None else init_v
    (Lx1, Lv1, px1), (Lx2, Lv2, px2) = dynamics.both(x, init_v, vb, log_jac=log_jac)   # one launch, both directions
    Lx, Lvm, px = torch.empty_like(x), torch.empty_like(x), torch.empty_like(px1)
    out = torch.empty_like(x) if do_mh_step else None
    if do_mh_step and u is None:
        u = _uniform(dynamics, B)
    _lib.check(_lib.lib().l2hmc_mix_accept(
        x.data_ptr(), Lx1.data_ptr(), Lv1.data_ptr(), px1.data_ptr(), Lx2.data_ptr(), Lv2.data_ptr(),
        px2.data_ptr(), mask.data_ptr(), None if u is None else _lib.as_dev(u, dynamics._device).data_ptr(),
        0, B, x.shape[1], Lx.data_ptr(), Lvm.data_ptr(), px.data_ptr(), None if out is None else out.data_ptr(),
        _lib.stream_ptr(x.device)))
    Lv = Lvm if init_v is not None else None       # :43-45 (quirk Q6)
    outputs = [out] if do_mh_step else []
    return Lx, Lv, px, outputs
