"""Host-side mirror of the reference's S/T/Q network classes.

Same constructor keywords, layer attribute names and call convention as
  l2hmc/network/generic_net.py:20-161 (GenericNet, _custom_dense)
  l2hmc/utils/network.py:89-114, 359-454 (`network` factory, Linear, ScaleTanh)
Weights are stored in the reference's layout (Dense kernel = [in, out]) on the
device; `pack()` produces the k-contiguous buffers the HIP kernels read
(include/l2hmc_hip.h, struct l2hmc_dense_net)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _trunc_normal(rng, shape, std):
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


class Dense:
    """tf.keras.layers.Dense stand-in: `kernel` [in, out], `bias` [out]
    (generic_net.py:149-161: variance_scaling(factor*2, FAN_IN, truncated normal), zero bias)."""

    def __init__(self, fan_in, units, factor=1., name=None, rng=None, device=None):
        rng = rng if rng is not None else np.random
        std = np.sqrt(1.3 * (factor * 2.) / fan_in)
        self.name = name
        self.kernel = torch.from_numpy(_trunc_normal(rng, (fan_in, units), std)).to(device)
        self.bias = torch.zeros(units, dtype=torch.float32, device=device)

    @property
    def variables(self):
        return [self.kernel, self.bias]


class _DenseSTQ:
    """Shared machinery: three input layers, one hidden layer, three heads."""
    q_tanh = 0
    _layer_names = ()   # (first-input, second-input, time, hidden, S, T, Q)
    _coeff_names = ()

    def _layers(self):
        return [getattr(self, n) for n in self._layer_names]

    @property
    def variables(self):
        self.sync_reference_layout()     # a trainer may have moved the flat master copy since the last read
        out = []
        for layer in self._layers():
            out.extend(layer.variables)
        out.extend(getattr(self, n) for n in self._coeff_names)
        return out

    trainable_variables = variables

    def state_dict(self):
        """Reference-layout tensors by name.  Once a trainer owns the flat master copy (flat_params), the
        optimiser only moves that buffer: every reader of the reference layout (this, `variables`,
        `save_weights` -- the call gauge_model.py:549-554 makes after training) refreshes it first."""
        self.sync_reference_layout()
        d = {}
        for n in self._layer_names:
            d[n + "/W"] = getattr(self, n).kernel
            d[n + "/b"] = getattr(self, n).bias
        for n in self._coeff_names:
            d[n] = getattr(self, n)
        return d

    def load_state(self, state):
        """Set weights from {name: array}; keys as state_dict() (the oracle uses the same)."""
        dev = self._device
        for n in self._layer_names:
            layer = getattr(self, n)
            W = _lib.as_dev(state[n + "/W"], dev)
            b = _lib.as_dev(state[n + "/b"], dev)
            if W.shape != layer.kernel.shape or b.shape != layer.bias.shape:
                raise ValueError(f"{n}: shape {tuple(W.shape)} != {tuple(layer.kernel.shape)}")
            layer.kernel, layer.bias = W, b
        for n in self._coeff_names:
            c = _lib.as_dev(state[n], dev).reshape(1, -1)
            if c.shape != getattr(self, n).shape:
                raise ValueError(f"{n}: bad shape {tuple(c.shape)}")
            setattr(self, n, c)
        self._packed = None
        self._flat = None

    def save_weights(self, path):
        """gauge_model.py:549-554 calls .save_weights on position_fn / momentum_fn."""
        np.savez(path, **{k: v.detach().cpu().numpy() for k, v in self.state_dict().items()})

    def load_weights(self, path):
        with np.load(path if str(path).endswith(".npz") else str(path) + ".npz") as f:
            self.load_state({k: f[k] for k in f.files})

    # ---- packing for the HIP kernels
    def _pack_tensors(self):
        """Reference-layout weights -> the k-contiguous buffers of struct l2hmc_dense_net."""
        la, lb, lt, lh, ls, ltr, lq = self._layers()
        cs, cq = (getattr(self, n) for n in self._coeff_names)
        return dict(
            w1_t=torch.cat([la.kernel, lb.kernel], dim=0).t().contiguous(),         # [H][Ka+Kb]
            wt=lt.kernel.contiguous(),                                               # [2][H]
            b1=(la.bias + lb.bias + lt.bias).contiguous(),
            wh_t=lh.kernel.t().contiguous(),                                          # [H][H] (out, in)
            bh=lh.bias.contiguous(),
            whd_t=torch.stack([ls.kernel.t(), ltr.kernel.t(), lq.kernel.t()]).contiguous(),  # [3][D][H]
            bhd=torch.stack([ls.bias, ltr.bias, lq.bias]).contiguous(),               # [3][D]
            coeff_s=cs.reshape(-1).contiguous(), coeff_q=cq.reshape(-1).contiguous())

    SEGMENTS = ("w1_t", "wt", "b1", "wh_t", "bh", "whd_t", "bhd", "coeff_s", "coeff_q")

    def flat_params(self):
        """Training master copy: ONE flat fp32 buffer [w1_t | wt | b1 | wh_t | bh | whd_t | bhd | coeff_s |
        coeff_q] whose slices are the buffers struct l2hmc_dense_net points at, so the optimiser is one
        element-wise pass and the data-parallel all-reduce one bucket.  Returns (flat, views, offsets); after
        this call pack() reads the flat buffer, and the reference-layout layer tensors are refreshed by
        sync_reference_layout()."""
        if getattr(self, "_flat", None) is None:
            bufs = self._pack_tensors()
            extra = self._extra_flat_tensors()
            bufs.update(extra)
            order = self.SEGMENTS + tuple(extra)
            flat = torch.cat([bufs[k].reshape(-1) for k in order])
            views, offsets, off = {}, {}, 0
            for k in order:
                n = bufs[k].numel()
                views[k] = flat[off:off + n].view(bufs[k].shape)
                offsets[k] = (off, off + n)
                off += n
            self._flat = (flat, views, offsets)
            self._packed = None
            self._front = None
        return self._flat

    def _extra_flat_tensors(self):
        """Further trainable tensors appended to the flat buffer (ConvNet3D: the Conv3D kernels / biases)."""
        return {}

    def refresh_packed(self):
        """After the flat buffer changed: rebuild the fragment-ordered image of the fused kernel."""
        if self._packed is not None and "packed" in self._packed[1]:
            st, bufs = self._packed
            _lib.check(_lib.lib().l2hmc_dense_pack(C.byref(st), bufs["packed"].data_ptr(), _lib.stream_ptr(self._device)))

    def sync_reference_layout(self):
        """Flat training buffer -> reference-layout layer tensors (state_dict / save_weights read those).
        The three first-layer biases share one packed bias and identical gradients; each takes a third of
        the packed bias's movement, which is what Adam does to them in the reference."""
        if getattr(self, "_flat", None) is None:
            return
        v = self._flat[1]
        la, lb, lt, lh, ls, ltr, lq = self._layers()
        Ka = la.kernel.shape[0]
        delta = (v["b1"] - (la.bias + lb.bias + lt.bias)) / 3.
        la.bias, lb.bias, lt.bias = la.bias + delta, lb.bias + delta, lt.bias + delta
        la.kernel = v["w1_t"][:, :Ka].t().contiguous()
        lb.kernel = v["w1_t"][:, Ka:].t().contiguous()
        lt.kernel = v["wt"].clone()
        lh.kernel, lh.bias = v["wh_t"].t().contiguous(), v["bh"].clone()
        for i, layer in enumerate((ls, ltr, lq)):
            layer.kernel, layer.bias = v["whd_t"][i].t().contiguous(), v["bhd"][i].clone()
        cs, cq = self._coeff_names
        setattr(self, cs, v["coeff_s"].reshape(1, -1).clone())
        setattr(self, cq, v["coeff_q"].reshape(1, -1).clone())

    def pack(self):
        """struct l2hmc_dense_net over device buffers (kept alive here); rebuilt after load_state()."""
        if self._packed is None:
            la, lb, lt, lh, ls, ltr, lq = self._layers()
            Ka, Kb, H, D = la.kernel.shape[0], lb.kernel.shape[0], lh.kernel.shape[0], ls.kernel.shape[1]
            bufs = ({k: self._flat[1][k] for k in self.SEGMENTS} if getattr(self, "_flat", None) is not None
                    else self._pack_tensors())
            st = _lib.DenseNet(D=D, H=H, Ka=Ka, Kb=Kb, q_tanh=self.q_tanh, reserved=0, packed=None,
                               **{k: _lib.dev_ptr(v, name=k) for k, v in bufs.items()})
            L = _lib.lib()
            nbytes = L.l2hmc_dense_pack_bytes(C.byref(st))
            if nbytes:      # fragment-ordered image for the fused whole-trajectory kernel
                bufs["packed"] = torch.empty(nbytes // 4, dtype=torch.float32, device=self._device)
                _lib.check(L.l2hmc_dense_pack(C.byref(st), bufs["packed"].data_ptr(), _lib.stream_ptr(self._device)))
                st.packed = bufs["packed"].data_ptr()
            self._packed = (st, bufs)
        return self._packed[0]

    def flat_tensors(self):
        """The packed device buffers in the field order of struct l2hmc_dense_net (w1_t, wt, b1, wh_t, bh, whd_t, bhd,
        coeff_s, coeff_q): what torch.ops.l2hmc.stq_dense takes as `weights` (l2hmc_amd/torch_ops.py)."""
        self.pack()
        bufs = self._packed[1]
        return [bufs[k] for k in ("w1_t", "wt", "b1", "wh_t", "bh", "whd_t", "bhd", "coeff_s", "coeff_q")]

    # ---- standalone evaluation: (S, T, Q) = net([a, b, t])
    def __call__(self, inputs):
        a, b, t = inputs[0], inputs[1], inputs[2]
        a = _lib.as_dev(a, self._device).reshape(a.shape[0], -1)
        b = _lib.as_dev(b, self._device).reshape(b.shape[0], -1)
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 2)
        tc, ts = float(t[0, 0]), float(t[0, 1])
        st = self.pack()
        rows = a.shape[0]
        S, T, Q = (torch.empty(rows, st.D, dtype=torch.float32, device=a.device) for _ in range(3))
        L = _lib.lib()
        ws, nb = self._ws.get(L.l2hmc_stq_ws_bytes(rows, st.H), a.device)
        _lib.check(L.l2hmc_stq_dense(C.byref(st), _lib.dev_ptr(a, name="a"), _lib.dev_ptr(b, name="b"), None,
                                     tc, ts, rows, S.data_ptr(), T.data_ptr(), Q.data_ptr(), ws, nb,
                                     _lib.stream_ptr(self._device)))
        return S, T, Q

    call = __call__


class GenericNet(_DenseSTQ):
    """generic_net.py:20-146.  kwargs: x_dim, num_hidden, factor, name_scope, links_shape."""
    q_tanh = 0
    _layer_names = ("v_layer", "x_layer", "t_layer", "h_layer", "scale_layer", "translation_layer",
                    "transformation_layer")
    _coeff_names = ("coeff_scale", "coeff_transformation")

    def __init__(self, model_name='GenericNet', rng=None, device=None, **kwargs):
        self.name = model_name
        for key, val in kwargs.items():
            setattr(self, key, val)
        if getattr(self, "name_scope", None) is None:
            self.name_scope = model_name
        self._device = device or torch.device("cuda", torch.cuda.current_device())
        dev, D, H = self._device, int(self.x_dim), int(self.num_hidden)
        # creation order follows generic_net.py:39-90 (it fixes the RNG stream order)
        self.x_layer = Dense(D, H, self.factor / 3., 'x_layer', rng, dev)
        self.v_layer = Dense(D, H, 1. / 3., 'v_layer', rng, dev)
        self.t_layer = Dense(2, H, 1. / 3., 't_layer', rng, dev)
        self.h_layer = Dense(H, H, 1., 'h_layer', rng, dev)
        self.scale_layer = Dense(H, D, 0.001, 'scale_layer', rng, dev)
        self.coeff_scale = torch.zeros(1, D, dtype=torch.float32, device=dev)
        self.translation_layer = Dense(H, D, 0.001, 'translation_layer', rng, dev)
        self.transformation_layer = Dense(H, D, 0.001, 'transformation_layer', rng, dev)
        self.coeff_transformation = torch.zeros(1, D, dtype=torch.float32, device=dev)
        self._packed = None
        self._ws = _lib.Workspace()


class Conv3D:
    """tf.keras.layers.Conv3D stand-in (parameters only): `kernel` [k0, k1, k2, Cin, Cout] with the
    Keras default glorot_uniform init, zero `bias` (conv_net.py:90-99 passes no initialiser)."""

    def __init__(self, kernel_size, cin, filters, name=None, rng=None, device=None):
        rng = rng if rng is not None else np.random
        shape = tuple(kernel_size) + (cin, filters)
        receptive = int(np.prod(kernel_size))
        lim = np.sqrt(6.0 / (cin * receptive + filters * receptive))
        self.name = name
        self.kernel = torch.from_numpy(rng.uniform(-lim, lim, shape).astype(np.float32)).to(device)
        self.bias = torch.zeros(filters, dtype=torch.float32, device=device)

    @property
    def variables(self):
        return [self.kernel, self.bias]


class ConvNet3D(_DenseSTQ):
    """conv_net.py:57-310, channels_last.  kwargs as the reference passes them
    (gauge_dynamics.py:123-134): _input_shape, links_shape, x_dim, factor, spatial_size, num_hidden,
    num_filters, filter_sizes, name_scope, data_format."""
    q_tanh = 0
    _layer_names = GenericNet._layer_names
    _coeff_names = GenericNet._coeff_names
    _conv_names = ("conv_x1", "conv_v1", "conv_x2", "conv_v2")

    def __init__(self, model_name, rng=None, device=None, **kwargs):
        self.name = model_name
        self.data_format = 'channels_last'
        self.filter_sizes = [(3, 3, 2), (2, 2, 2)]
        for key, val in kwargs.items():
            setattr(self, key, val)
        if self.data_format != 'channels_last':
            raise NotImplementedError("only data_format='channels_last' (the CPU/TF reference layout) is built; "
                                      "channels_first reinterprets the link buffer (SURVEY.md 8a)")
        if [tuple(f) for f in self.filter_sizes] != [(3, 3, 2), (2, 2, 2)]:
            raise NotImplementedError("the conv kernels are written for filter_sizes [(3,3,2),(2,2,2)]")
        self._device = device or torch.device("cuda", torch.cuda.current_device())
        dev, D, H, F = self._device, int(self.x_dim), int(self.num_hidden), int(self.num_filters)
        T, X = int(self.links_shape[0]), int(self.links_shape[1])
        if T % 4 or X % 4:
            raise ValueError("ConvNet3D on the HIP path needs lattice extents that are multiples of 4")
        self.nflat = (T // 4) * (X // 4) * 2 * F
        self.coeff_scale = torch.zeros(1, D, dtype=torch.float32, device=dev)
        self.coeff_transformation = torch.zeros(1, D, dtype=torch.float32, device=dev)
        # creation order of conv_net.py:90-164
        self.conv_x1 = Conv3D((3, 3, 2), 1, F, 'conv_x1', rng, dev)
        self.conv_v1 = Conv3D((3, 3, 2), 1, F, 'conv_v1', rng, dev)
        self.conv_x2 = Conv3D((2, 2, 2), F, 2 * F, 'conv_x2', rng, dev)
        self.conv_v2 = Conv3D((2, 2, 2), F, 2 * F, 'conv_v2', rng, dev)
        self.x_layer = Dense(self.nflat, H, self.factor / 3., 'x_layer', rng, dev)
        self.v_layer = Dense(self.nflat, H, 1. / 3., 'v_layer', rng, dev)
        self.t_layer = Dense(2, H, 1. / 3., 't_layer', rng, dev)
        self.h_layer = Dense(H, H, 1., 'h_layer', rng, dev)
        self.scale_layer = Dense(H, D, 0.001, 'scale_layer', rng, dev)
        self.translation_layer = Dense(H, D, 0.001, 'translation_layer', rng, dev)
        self.transformation_layer = Dense(H, D, 0.001, 'transformation_layer', rng, dev)
        self._packed = None
        self._front = None
        self._ws = _lib.Workspace()

    @property
    def variables(self):
        self.sync_reference_layout()
        out = [self.coeff_scale, self.coeff_transformation]
        for n in self._conv_names:
            out.extend(getattr(self, n).variables)
        for layer in self._layers():
            out.extend(layer.variables)
        return out

    trainable_variables = variables

    def state_dict(self):
        d = super().state_dict()
        for n in self._conv_names:
            d[n + "/W"] = getattr(self, n).kernel
            d[n + "/b"] = getattr(self, n).bias
        return d

    def load_state(self, state):
        super().load_state(state)
        for n in self._conv_names:
            layer = getattr(self, n)
            W = _lib.as_dev(state[n + "/W"], self._device)
            b = _lib.as_dev(state[n + "/b"], self._device)
            if W.shape != layer.kernel.shape or b.shape != layer.bias.shape:
                raise ValueError(f"{n}: shape {tuple(W.shape)} != {tuple(layer.kernel.shape)}")
            layer.kernel, layer.bias = W, b
        self._front = None

    # struct l2hmc_conv3d_front field -> (layer, attribute): *_a = first input (conv_v*), *_b = second (conv_x*)
    _FRONT = {"w1_a": ("conv_v1", "kernel"), "b1_a": ("conv_v1", "bias"), "w2_a": ("conv_v2", "kernel"),
              "b2_a": ("conv_v2", "bias"), "w1_b": ("conv_x1", "kernel"), "b1_b": ("conv_x1", "bias"),
              "w2_b": ("conv_x2", "kernel"), "b2_b": ("conv_x2", "bias")}

    def _extra_flat_tensors(self):
        return {k: getattr(getattr(self, layer), attr).contiguous() for k, (layer, attr) in self._FRONT.items()}

    def pack_front(self):
        """struct l2hmc_conv3d_front over the Keras-layout kernels (or their slices of the flat training buffer)."""
        if self._front is None:
            flat = getattr(self, "_flat", None)
            bufs = ({k: flat[1][k] for k in self._FRONT} if flat is not None else self._extra_flat_tensors())
            self._front = _lib.Conv3DFront(F=int(self.num_filters), reserved=0,
                                           **{k: _lib.dev_ptr(v, name=k) for k, v in bufs.items()})
            self._front_bufs = bufs          # keep the tensors alive
        return self._front

    def sync_reference_layout(self):
        super().sync_reference_layout()
        if getattr(self, "_flat", None) is not None:
            for k, (layer, attr) in self._FRONT.items():
                setattr(getattr(self, layer), attr, self._flat[1][k].clone())

    def __call__(self, inputs):
        """conv_net.py:247-280: (scale, translation, transformation) = net([v, x, t])."""
        a, b, t = inputs[0], inputs[1], inputs[2]
        a = _lib.as_dev(a, self._device).reshape(a.shape[0], -1)
        b = _lib.as_dev(b, self._device).reshape(b.shape[0], -1)
        t = torch.as_tensor(t, dtype=torch.float32).reshape(-1, 2)
        st, fr = self.pack(), self.pack_front()
        T, X = int(self.links_shape[0]), int(self.links_shape[1])
        rows = a.shape[0]
        S, Tr, Q = (torch.empty(rows, st.D, dtype=torch.float32, device=a.device) for _ in range(3))
        L = _lib.lib()
        ws, nb = self._ws.get(L.l2hmc_stq_conv3d_ws_bytes(rows, st.H, T, X, fr.F), a.device)
        _lib.check(L.l2hmc_stq_conv3d(C.byref(fr), C.byref(st), T, X, _lib.dev_ptr(a, name="a"),
                                      _lib.dev_ptr(b, name="b"), None, float(t[0, 0]), float(t[0, 1]), rows,
                                      S.data_ptr(), Tr.data_ptr(), Q.data_ptr(), ws, nb, _lib.stream_ptr(self._device)))
        return S, Tr, Q

    call = __call__


class MLPNet(_DenseSTQ):
    """utils/network.py:89-114: what `network(x_dim, scope, factor, num_nodes)` returns.
    Callable on [a, b, t, aux]; aux is ignored (the `lambda _: 0.` slot)."""
    q_tanh = 1
    _layer_names = ("embed_1", "embed_2", "embed_3", "linear_1", "linear_s", "linear_t", "linear_f")
    _coeff_names = ("scale_s", "scale_f")

    def __init__(self, x_dim, scope, factor, num_nodes=50, rng=None, device=None):
        self.name = self.scope = scope
        self.x_dim, self.num_nodes, self.factor = int(x_dim), int(num_nodes), factor
        self._device = device or torch.device("cuda", torch.cuda.current_device())
        dev, D, H = self._device, self.x_dim, self.num_nodes
        self.embed_1 = Dense(D, H, 1. / 3., 'embed_1', rng, dev)
        self.embed_2 = Dense(D, H, factor / 3., 'embed_2', rng, dev)
        self.embed_3 = Dense(2, H, 1. / 3., 'embed_3', rng, dev)
        self.linear_1 = Dense(H, H, 1., 'linear_1', rng, dev)
        self.linear_s = Dense(H, D, 0.001, 'linear_s', rng, dev)
        self.scale_s = torch.zeros(1, D, dtype=torch.float32, device=dev)
        self.linear_t = Dense(H, D, 0.001, 'linear_t', rng, dev)
        self.linear_f = Dense(H, D, 0.001, 'linear_f', rng, dev)
        self.scale_f = torch.zeros(1, D, dtype=torch.float32, device=dev)
        self._packed = None
        self._ws = _lib.Workspace()
    # (callable on [a, b, t(, aux)] through l2hmc_stq_dense like the other nets -- any width: the layer-by-layer path of
    #  l2hmc_amd.dynamics.Dynamics; the one-launch toy kernels evaluate the same weights inside their trajectory)


def network(x_dim, scope, factor, num_nodes=50, rng=None, device=None):
    """utils/network.py:89 -- the `net_factory` callers hand to Dynamics."""
    return MLPNet(x_dim, scope, factor, num_nodes, rng=rng, device=device)
