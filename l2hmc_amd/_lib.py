"""ctypes binding of libl2hmc_hip.so (include/l2hmc_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not a
contiguous fp32 CUDA tensor, the call raises.  PyTorch-ROCm is used only for
device memory and streams."""
import ctypes as C
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# L2HMC_LIB_PATH: load another build of the same ABI (diagnostic builds under tools/_diag); default = the in-tree library
LIB_PATH = os.environ.get("L2HMC_LIB_PATH") or os.path.join(_HERE, "libl2hmc_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "l2hmc_hip.h")

c_float_p = C.c_void_p   # device pointers travel as integers
MAX_MIX, MAX_SMALL_DIM, MAX_SMALL_NODES = 8, 8, 64
PLAN_LAYERED, PLAN_CONV3D, PLAN_SELECTED_ONLY, PLAN_RECOMPUTE, PLAN_TILES16_ONLY, PLAN_ALL_COLUMNS = 1, 2, 4, 8, 16, 32
GRAD_BUCKET_REST = 6
BUCKET_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int32)


class DenseNet(C.Structure):
    _fields_ = [("D", C.c_int32), ("H", C.c_int32), ("Ka", C.c_int32), ("Kb", C.c_int32),
                ("w1_t", c_float_p), ("wt", c_float_p), ("b1", c_float_p), ("wh_t", c_float_p),
                ("bh", c_float_p), ("whd_t", c_float_p), ("bhd", c_float_p),
                ("coeff_s", c_float_p), ("coeff_q", c_float_p),
                ("q_tanh", C.c_int32), ("reserved", C.c_int32), ("packed", c_float_p)]


class DenseGrads(C.Structure):
    _fields_ = [(n, c_float_p) for n in ("w1_t", "wt", "b1", "wh_t", "bh", "whd_t", "bhd", "coeff_s", "coeff_q")]


class Conv3DGrads(C.Structure):
    _fields_ = [(n, c_float_p) for n in ("w1_a", "b1_a", "w2_a", "b2_a", "w1_b", "b1_b", "w2_b", "b2_b")]


class Conv3DFront(C.Structure):
    _fields_ = [("F", C.c_int32), ("reserved", C.c_int32),
                ("w1_a", c_float_p), ("b1_a", c_float_p), ("w2_a", c_float_p), ("b2_a", c_float_p),
                ("w1_b", c_float_p), ("b1_b", c_float_p), ("w2_b", c_float_p), ("b2_b", c_float_p)]


class GaugePlan(C.Structure):
    _fields_ = [("T", C.c_int32), ("X", C.c_int32), ("num_steps", C.c_int32), ("hmc", C.c_int32),
                ("eps", C.c_float), ("flags", C.c_int32), ("masks", c_float_p),
                ("xnet", DenseNet), ("vnet", DenseNet), ("xfront", Conv3DFront), ("vfront", Conv3DFront)]


class MogTarget(C.Structure):
    _fields_ = [("dim", C.c_int32), ("K", C.c_int32), ("is_gaussian", C.c_int32),
                ("temperature", C.c_float), ("mu", c_float_p), ("prec", c_float_p),
                ("log_const", c_float_p)]


class SmallPlan(C.Structure):
    _fields_ = [("x_dim", C.c_int32), ("num_nodes", C.c_int32), ("trajectory_length", C.c_int32),
                ("hmc", C.c_int32), ("eps", C.c_float), ("first_layer_form", C.c_int32), ("masks", c_float_p),
                ("xnet", DenseNet), ("vnet", DenseNet), ("target", MogTarget)]


_P, _I32, _I64, _F, _SZ, _U64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t, C.c_uint64
_PROTOS = {
    "l2hmc_abi_version": (C.c_int, []),
    "l2hmc_last_error": (C.c_char_p, []),
    "l2hmc_u1_action_force": (C.c_int, [_P, _I64, _I32, _I32, _F, _P, _P, _P, _P, _P]),
    "l2hmc_u1_plaq_sums": (C.c_int, [_P, _I64, _I32, _I32, _P, _P]),
    "l2hmc_wrap_angle": (C.c_int, [_P, _I64, _P, _P]),
    "l2hmc_kinetic_energy": (C.c_int, [_P, _I64, _I32, _P, _P]),
    "l2hmc_stq_ws_bytes": (_SZ, [_I64, _I32]),
    "l2hmc_stq_dense": (C.c_int, [C.POINTER(DenseNet), _P, _P, _P, _F, _F, _I64, _P, _P, _P, _P, _SZ, _P]),
    "l2hmc_stq_conv3d_ws_bytes": (_SZ, [_I64, _I32, _I32, _I32, _I32]),
    "l2hmc_stq_conv3d": (C.c_int, [C.POINTER(Conv3DFront), C.POINTER(DenseNet), _I32, _I32, _P, _P, _P, _F, _F,
                                   _I64, _P, _P, _P, _P, _SZ, _P]),
    "l2hmc_dense_pack_bytes": (_SZ, [C.POINTER(DenseNet)]),
    "l2hmc_dense_pack": (C.c_int, [C.POINTER(DenseNet), _P, _P]),
    "l2hmc_lf_update_v": (C.c_int, [_P, _P, _P, _P, _P, _F, _I32, _I64, _I32, _P, _P, _P]),
    "l2hmc_lf_update_x": (C.c_int, [_P, _P, _P, _P, _P, _P, _F, _I32, _I64, _I32, _P, _P, _P]),
    "l2hmc_accept_prob": (C.c_int, [_P, _P, _P, _I64, _P, _P]),
    "l2hmc_mix_accept": (C.c_int, [_P] * 9 + [_I32, _I64, _I32, _P, _P, _P, _P, _P]),
    "l2hmc_gauge_ws_bytes": (_SZ, [C.POINTER(GaugePlan), _I64]),
    "l2hmc_gauge_plan_fused": (C.c_int, [C.POINTER(GaugePlan)]),
    "l2hmc_gauge_step_plan": (C.c_int, [_I64, _I32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "l2hmc_gauge_leapfrog": (C.c_int, [C.POINTER(GaugePlan), _F, _I32, _P, _P, _P, _I64, _P, _P, _SZ, _P]),
    "l2hmc_gauge_trajectory": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _P, _P, _I64, _P, _P, _P, _P, _P,
                                         _SZ, _P]),
    "l2hmc_gauge_transition_ws_bytes": (_SZ, [C.POINTER(GaugePlan), _I64, _I32]),
    "l2hmc_gauge_transition": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _P, _P, _P, _P, _I64, _I32, _P, _P,
                                         _P, _P, _P, _SZ, _P]),
    "l2hmc_gauge_transition_draw": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _I64, _U64, _U64, _P, _P, _P, _P, _P, _SZ,
                                              _P]),
    "l2hmc_gauge_mcmc_step_ws_bytes": (_SZ, [C.POINTER(GaugePlan), _I64]),
    "l2hmc_gauge_mcmc_step": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _I64, _U64, _U64, _P, _P, _P, _P, _P, _P,
                                        _SZ, _P]),
    "l2hmc_gauge_mcmc_step_ex": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _P, _I64, _U64, _U64, _P, _P, _P, _P, _P, _P, _P,
                                           _SZ, _P]),
    "l2hmc_gauge_loss_terms": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _F, _F, _F, _F, _P, _P]),
    "l2hmc_gauge_train_ws_bytes": (_SZ, [C.POINTER(GaugePlan), _I64]),
    "l2hmc_gauge_train_forward": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _P, _P, _I64, _P, _P, _P, _P, _P, _SZ,
                                            _P]),
    "l2hmc_gauge_train_backward": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _I64, _P, _P, _P,
                                             C.POINTER(DenseGrads), C.POINTER(DenseGrads),
                                             C.POINTER(Conv3DGrads), C.POINTER(Conv3DGrads), _P, _P, _SZ, _P]),
    "l2hmc_gauge_train_backward_buckets": (C.c_int, [C.POINTER(GaugePlan), _F, _P, _I64, _P, _P, _P,
                                                     C.POINTER(DenseGrads), C.POINTER(DenseGrads),
                                                     C.POINTER(Conv3DGrads), C.POINTER(Conv3DGrads), _P, _P, _SZ, _P,
                                                     BUCKET_FN, _P]),
    "l2hmc_gauge_loss_backward": (C.c_int, [_I32, _I32, _F, _P, _P, _P, _P, _I64, _I32, _F, _F, _F, _F, _F, _P, _P,
                                            _P, _P, _P]),
    "l2hmc_grad_sumsq": (C.c_int, [_P, _I64, _I64, _I64, _P, _I32, _P]),
    "l2hmc_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _P, _F, _I64, _I64, _P]),
    "l2hmc_mog_energy_grad": (C.c_int, [C.POINTER(MogTarget), _P, _I64, _P, _P, _P]),
    "l2hmc_small_trajectory": (C.c_int, [C.POINTER(SmallPlan), _P, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "l2hmc_small_propose": (C.c_int, [C.POINTER(SmallPlan), _P, _I64, C.c_uint64, C.c_uint64, _P, _P, _P, _P, _P]),
    "l2hmc_small_train_ws_bytes": (_SZ, [C.POINTER(SmallPlan), _I64]),
    "l2hmc_small_train_step": (C.c_int, [C.POINTER(SmallPlan), _P, _P, _P, _I64, _F, _F, _P, _P, _P, _P, _P, _P, _SZ,
                                         _P]),
    "l2hmc_profile_begin": (C.c_int, [_I32]),
    "l2hmc_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "l2hmc_fill_normal": (C.c_int, [_P, _I64, _U64, _U64, _P]),
    "l2hmc_fill_uniform": (C.c_int, [_P, _I64, _U64, _U64, _P]),
}

_lib = None


def declared_symbols():
    """Every function name include/l2hmc_hip.h declares."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(l2hmc_[a-z0-9_]+)\s*\(", text)))


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m l2hmc_amd.build` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.l2hmc_abi_version() != 1:
            raise RuntimeError("libl2hmc_hip.so ABI version mismatch")
        _lib = handle
    return _lib


class L2HMCError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        msg = lib().l2hmc_last_error().decode()
        kind = {1: "bad argument", 2: "HIP error", 3: "workspace too small"}.get(rc, f"error {rc}")
        if rc == 1:
            raise ValueError(f"l2hmc_hip: {kind}: {msg}")
        raise L2HMCError(f"l2hmc_hip: {kind}: {msg}")


def dev_ptr(t, dtype=torch.float32, name="tensor"):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a CUDA (ROCm) tensor; the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t.data_ptr()


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device` (default: the current device).  Kernels are launched
    on the CURRENT device, so a tensor living elsewhere is an error, not a silent cross-device launch."""
    if device is not None:
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the HIP path has no CPU fallback")
        idx = torch.cuda.current_device() if device.index is None else device.index
        if idx != torch.cuda.current_device():
            raise RuntimeError(f"object lives on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}; "
                               f"wrap the call in `with torch.cuda.device({idx}):`")
        return torch.cuda.current_stream(idx).cuda_stream
    return torch.cuda.current_stream().cuda_stream


def step_draw_index(draws):
    """The native MCMC step (l2hmc_gauge_mcmc_step) consumes the Philox stream pair (2d, 2d+1) of its `draw`
    argument; `GaugeDynamics._normal/_uniform` consume stream `_draws` and advance it by one.  ONE counter serves
    both: the step takes the next even-aligned pair at or after `draws`.  Returns (d, new_draws)."""
    d = (int(draws) + 1) // 2
    return d, 2 * d + 2


def as_dev(a, device=None, dtype=torch.float32):
    """numpy / tensor -> contiguous CUDA tensor of `dtype` (copy only if needed)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    if not isinstance(a, torch.Tensor):
        a = torch.as_tensor(a)
    return a.to(device=device, dtype=dtype).contiguous()


class Workspace:
    """Grow-only scratch buffer handed to the C ABI (which never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        return self.buf.data_ptr(), self.buf.numel()
