"""Oracle (test infrastructure): the scale/translate/transform (S/T/Q) networks.

NumPy restatement of
  l2hmc/network/generic_net.py:20-161    GenericNet (+ _custom_dense init)
  l2hmc/network/conv_net.py:57-310,502-514  ConvNet3D (channels_last)
  l2hmc/utils/network.py:89-114,359-454  `network` MLP (Linear/Zip/Parallel/ScaleTanh)

Third-party behaviour restated explicitly (TensorFlow 1.x / Keras, absent from
/root/reference; SURVEY.md 8a "Layout conventions"):
  * Dense: y = x @ W + b, W is [in, out].
  * Conv3D channels_last: kernel [kd0, kd1, kd2, Cin, Cout], cross-correlation,
    padding='same' => zeros, total pad k-1 with floor((k-1)/2) before.
  * MaxPool3D(2, strides 2, 'same'): out = ceil(in/2), max over in-bounds cells.
  * Flatten: row-major over (h, w, depth, channel).
  * variance_scaling_initializer(factor=2f, FAN_IN, uniform=False):
    truncated normal (|z| <= 2 sigma), sigma = sqrt(1.3 * 2f / fan_in).
"""
import numpy as np


# ----------------------------------------------------------------- init ----
def trunc_normal(rng, shape, std, dtype=np.float64):
    """tf.truncated_normal: redraw anything beyond two standard deviations."""
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(dtype)


def _dense_init(rng, fan_in, units, factor, bias_std=0.0, dtype=np.float64):
    """generic_net.py:149-161 `_custom_dense(units, factor)`."""
    std = np.sqrt(1.3 * (factor * 2.0) / fan_in)
    W = trunc_normal(rng, (fan_in, units), std, dtype)
    b = (rng.standard_normal(units) * bias_std).astype(dtype)
    return W, b


def init_generic_net(rng, x_dim, num_hidden, factor, head_factor=0.001,
                     bias_std=0.0, coeff_std=0.0, dtype=np.float64):
    """generic_net.py:36-90.  `head_factor`, `bias_std`, `coeff_std` default to
    the reference initialisation; the "stress" regime of SURVEY.md 8d raises
    them so that exp/tanh/log-det paths are exercised at O(0.3) magnitudes."""
    p = {}
    p['x_layer/W'], p['x_layer/b'] = _dense_init(rng, x_dim, num_hidden, factor / 3., bias_std, dtype)
    p['v_layer/W'], p['v_layer/b'] = _dense_init(rng, x_dim, num_hidden, 1. / 3., bias_std, dtype)
    p['t_layer/W'], p['t_layer/b'] = _dense_init(rng, 2, num_hidden, 1. / 3., bias_std, dtype)
    p['h_layer/W'], p['h_layer/b'] = _dense_init(rng, num_hidden, num_hidden, 1., bias_std, dtype)
    for name in ('scale_layer', 'translation_layer', 'transformation_layer'):
        p[name + '/W'], p[name + '/b'] = _dense_init(rng, num_hidden, x_dim, head_factor, bias_std, dtype)
    p['coeff_scale'] = (rng.standard_normal((1, x_dim)) * coeff_std).astype(dtype)
    p['coeff_transformation'] = (rng.standard_normal((1, x_dim)) * coeff_std).astype(dtype)
    return p


def _glorot_uniform(rng, shape, dtype):
    """Keras default Conv3D kernel_initializer (conv_net.py:91-99 sets none)."""
    receptive = int(np.prod(shape[:-2]))
    fan_in, fan_out = shape[-2] * receptive, shape[-1] * receptive
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, shape).astype(dtype)


def conv3d_flat_size(L, num_filters):
    """Spatial extent after conv/pool x2 (conv_net.py:247-262): ceil(ceil(L/2)/2)^2 * 1 * 2F."""
    l2 = -(-(-(-L // 2)) // 2)
    return l2 * l2 * 2 * num_filters


def init_conv3d_net(rng, L, x_dim, num_hidden, num_filters, factor,
                    head_factor=0.001, bias_std=0.0, coeff_std=0.0, dtype=np.float64):
    """conv_net.py:57-207 (filter_sizes [(3,3,2),(2,2,2)], gauge_dynamics.py:121-143)."""
    F = num_filters
    p = {}
    for s in ('x', 'v'):
        p[f'conv_{s}1/W'] = _glorot_uniform(rng, (3, 3, 2, 1, F), dtype)
        p[f'conv_{s}1/b'] = (rng.standard_normal(F) * bias_std).astype(dtype)
        p[f'conv_{s}2/W'] = _glorot_uniform(rng, (2, 2, 2, F, 2 * F), dtype)
        p[f'conv_{s}2/b'] = (rng.standard_normal(2 * F) * bias_std).astype(dtype)
    nflat = conv3d_flat_size(L, F)
    p['x_layer/W'], p['x_layer/b'] = _dense_init(rng, nflat, num_hidden, factor / 3., bias_std, dtype)
    p['v_layer/W'], p['v_layer/b'] = _dense_init(rng, nflat, num_hidden, 1. / 3., bias_std, dtype)
    p['t_layer/W'], p['t_layer/b'] = _dense_init(rng, 2, num_hidden, 1. / 3., bias_std, dtype)
    p['h_layer/W'], p['h_layer/b'] = _dense_init(rng, num_hidden, num_hidden, 1., bias_std, dtype)
    for name in ('scale_layer', 'translation_layer', 'transformation_layer'):
        p[name + '/W'], p[name + '/b'] = _dense_init(rng, num_hidden, x_dim, head_factor, bias_std, dtype)
    p['coeff_scale'] = (rng.standard_normal((1, x_dim)) * coeff_std).astype(dtype)
    p['coeff_transformation'] = (rng.standard_normal((1, x_dim)) * coeff_std).astype(dtype)
    return p


def init_mlp_net(rng, x_dim, factor, num_nodes=50, head_factor=0.001,
                 bias_std=0.0, coeff_std=0.0, dtype=np.float64):
    """utils/network.py:89-114 `network(x_dim, scope, factor, num_nodes)`."""
    p = {}
    p['embed_1/W'], p['embed_1/b'] = _dense_init(rng, x_dim, num_nodes, 1. / 3., bias_std, dtype)
    p['embed_2/W'], p['embed_2/b'] = _dense_init(rng, x_dim, num_nodes, factor / 3., bias_std, dtype)
    p['embed_3/W'], p['embed_3/b'] = _dense_init(rng, 2, num_nodes, 1. / 3., bias_std, dtype)
    p['linear_1/W'], p['linear_1/b'] = _dense_init(rng, num_nodes, num_nodes, 1., bias_std, dtype)
    for name in ('linear_s', 'linear_t', 'linear_f'):
        p[name + '/W'], p[name + '/b'] = _dense_init(rng, num_nodes, x_dim, head_factor, bias_std, dtype)
    p['scale_s'] = (rng.standard_normal((1, x_dim)) * coeff_std).astype(dtype)
    p['scale_f'] = (rng.standard_normal((1, x_dim)) * coeff_std).astype(dtype)
    return p


def cast_params(p, dtype):
    return {k: np.asarray(v, dtype=dtype) for k, v in p.items()}


# -------------------------------------------------------------- forward ----
def _relu(a):
    return np.maximum(a, 0)


def generic_net(p, inputs):
    """generic_net.py:129-146.  inputs = [v, x, t]: the FIRST entry goes through
    v_layer and the SECOND through x_layer (for VNet the caller passes
    [position, grad, t], gauge_dynamics.py:493-495).  No tanh on `transformation`
    (quirk Q1)."""
    v, x, t = inputs
    h = (v @ p['v_layer/W'] + p['v_layer/b']) + (x @ p['x_layer/W'] + p['x_layer/b']) \
        + (t @ p['t_layer/W'] + p['t_layer/b'])
    h = _relu(h)
    h = _relu(h @ p['h_layer/W'] + p['h_layer/b'])
    scale = np.tanh(h @ p['scale_layer/W'] + p['scale_layer/b']) * np.exp(p['coeff_scale'])
    translation = h @ p['translation_layer/W'] + p['translation_layer/b']
    transformation = (h @ p['transformation_layer/W'] + p['transformation_layer/b']) \
        * np.exp(p['coeff_transformation'])
    return scale, translation, transformation


def _same_pad(n, k):
    """TF 'same', stride 1: total k-1, floor((k-1)/2) before."""
    tot = k - 1
    return tot // 2, tot - tot // 2


def conv3d_same_relu(a, W, b):
    """Keras Conv3D(padding='same', activation=relu), channels_last.
    a: [B, d0, d1, d2, Cin], W: [k0, k1, k2, Cin, Cout]."""
    k0, k1, k2, cin, cout = W.shape
    pads = [(0, 0)] + [_same_pad(a.shape[i + 1], k) for i, k in enumerate((k0, k1, k2))] + [(0, 0)]
    ap = np.pad(a, pads)
    B, n0, n1, n2 = a.shape[:4]
    out = np.zeros((B, n0, n1, n2, cout), dtype=a.dtype)
    for i0 in range(k0):
        for i1 in range(k1):
            for i2 in range(k2):
                patch = ap[:, i0:i0 + n0, i1:i1 + n1, i2:i2 + n2, :]
                out = out + patch @ W[i0, i1, i2]
    return _relu(out + b)


def maxpool3d_same(a):
    """MaxPooling3D(pool 2, strides 2, 'same'): ceil(n/2) outputs per axis, the
    window clipped at the upper boundary (padding never wins the max)."""
    B = a.shape[0]
    n = a.shape[1:4]
    o = [-(-x // 2) for x in n]
    out = np.full((B, o[0], o[1], o[2], a.shape[4]), -np.inf, dtype=a.dtype)
    for i0 in range(2):
        for i1 in range(2):
            for i2 in range(2):
                sub = a[:, i0::2, i1::2, i2::2, :]
                s = sub.shape
                out[:, :s[1], :s[2], :s[3], :] = np.maximum(out[:, :s[1], :s[2], :s[3], :], sub)
    return out


def conv3d_front(p, a, which, links_shape):
    """conv_net.py:251-262: reshape_5D -> conv1 -> pool -> conv2 -> pool -> flatten."""
    T, X, d = links_shape
    a = a.reshape(a.shape[0], T, X, d, 1)           # conv_net.py:300-306
    a = maxpool3d_same(conv3d_same_relu(a, p[f'conv_{which}1/W'], p[f'conv_{which}1/b']))
    a = maxpool3d_same(conv3d_same_relu(a, p[f'conv_{which}2/W'], p[f'conv_{which}2/b']))
    return a.reshape(a.shape[0], -1)


def conv3d_net(p, inputs, links_shape):
    """conv_net.py:247-280 (channels_last)."""
    v, x, t = inputs
    v = conv3d_front(p, v, 'v', links_shape)
    x = conv3d_front(p, x, 'x', links_shape)
    return generic_net(p, [v, x, t])


def mlp_net(p, inputs):
    """utils/network.py:89-114: Zip(embed_1, embed_2, embed_3, 0) -> sum -> relu
    -> linear_1 -> relu -> Parallel(S = exp(scale_s) tanh(.), T, F = exp(scale_f) tanh(.)).
    tanh on BOTH S and F here, unlike GenericNet (quirk Q1)."""
    a, b, t = inputs[:3]
    h = (a @ p['embed_1/W'] + p['embed_1/b']) + (b @ p['embed_2/W'] + p['embed_2/b']) \
        + (t @ p['embed_3/W'] + p['embed_3/b']) + 0.
    h = _relu(h)
    h = _relu(h @ p['linear_1/W'] + p['linear_1/b'])
    S = np.exp(p['scale_s']) * np.tanh(h @ p['linear_s/W'] + p['linear_s/b'])
    T = h @ p['linear_t/W'] + p['linear_t/b']
    F = np.exp(p['scale_f']) * np.tanh(h @ p['linear_f/W'] + p['linear_f/b'])
    return S, T, F


def zero_net(inputs):
    """hmc=True: gauge_dynamics.py:102-108, utils/dynamics.py:75-78."""
    z = np.zeros_like(inputs[0])
    return z, z, z
