"""Oracle (test infrastructure): the training loss of the lattice model, forward value only.

NumPy restatement of l2hmc/gauge_model.py:728-797 `_calc_loss` with its helpers
  :632-657 `_create_metric_fn`      :94-108 `project_angle_approx`      :718-725 `_calc_top_charges_diff`
Quirk kept as written (Q9): BOTH auxiliary terms compare z with the proposal of x (`x_`), not with the
proposal `z_` of the auxiliary chain (gauge_model.py:775, :789); `z_` only enters through `pz`.
"""
import numpy as np

from . import lattice as lat

METRICS = {
    'l1': lambda a, b: np.abs(a - b),
    'l2': lambda a, b: np.square(a - b),
    'cos': lambda a, b: np.abs(np.cos(a) - np.cos(b)),
    'cos2': lambda a, b: np.square(np.cos(a) - np.cos(b)),
    'cos_diff': lambda a, b: 1. - np.cos(a - b),
}


def calc_loss_terms(x, x_prop, px, z, pz, T, X, metric='cos_diff', loss_scale=1., aux_weight=1., std_weight=1.,
                    charge_weight=1.):
    """Per-chain (std_loss + charge_loss); the scalar loss is its mean (gauge_model.py:795)."""
    eps = 1e-3
    m = METRICS[metric]
    x_std = np.sum(m(x, x_prop), axis=1) * px + eps
    z_std = aux_weight * (np.sum(m(z, x_prop), axis=1) * pz + eps)
    ls = loss_scale
    std_loss = std_weight * (ls * (1. / x_std + 1. / z_std) - (x_std + z_std) / ls)
    xq = px * lat.top_charge_diff(x, x_prop, T, X, fft=True) + eps
    zq = aux_weight * (pz * lat.top_charge_diff(z, x_prop, T, X, fft=True) + eps)
    return std_loss + charge_weight * (xq + zq)


def calc_loss(dynamics_oracle, x, beta, draws_x, z, draws_z, **weights):
    """`draws_*` = (v0_f, v0_b, coin, u) for the two apply_transition calls (:753, :758)."""
    x_prop, _, px, x_out = dynamics_oracle.apply_transition(x, beta, *draws_x)
    _, _, pz, _ = dynamics_oracle.apply_transition(z, beta, *draws_z)
    terms = calc_loss_terms(x, x_prop, px, z, pz, dynamics_oracle.T, dynamics_oracle.X, **weights)
    x_dq = lat.top_charge_diff(x, x_out, dynamics_oracle.T, dynamics_oracle.X, fft=False).astype(np.int32)
    return float(np.mean(terms)), x_out, px, x_dq, terms
