"""Oracle (test infrastructure): direct O(n^2) restatements of the chain statistics of
l2hmc/utils/autocorr.py:23-34 (autocorr_fast), :107-126 (autocorr_func_1d), :134-199 (integrated_time),
written as lag sums so that the FFT formulations in l2hmc_amd/stats.py have an independent check."""
import numpy as np


def acf_direct(x, unbiased=False):
    """sum_t (x_t - mean)(x_{t+k} - mean), optionally / (N - k); normalised to lag 0 = 1."""
    x = np.asarray(x, dtype=np.float64)
    x = x - x.mean()
    n = len(x)
    out = np.array([np.dot(x[: n - k], x[k:]) for k in range(n)])
    if unbiased:
        out = out / (n - np.arange(n))
    return out / out[0]


def integrated_time_direct(x, c=5):
    """x [steps, walkers]: walker-averaged acf, tau(m) = 2 cumsum - 1, first m >= c tau(m)."""
    x = np.asarray(x, dtype=np.float64)
    f = np.mean([acf_direct(x[:, k]) for k in range(x.shape[1])], axis=0)
    taus = 2. * np.cumsum(f) - 1.
    m = np.arange(len(taus)) < c * taus
    w = int(np.argmin(m)) if np.any(m) else len(taus) - 1
    return taus[w]


def autocovariance_direct(X, tau=0):
    """l2hmc/utils/func_utils.py:45-54: mean over t of sum_{chains,dims} x_t x_{t+tau} / n_chains (uncentred)."""
    X = np.asarray(X, dtype=np.float64)
    dT, dN, dX = X.shape
    s = 0.
    for t in range(dT - tau):
        x1, x2 = X[t], X[t + tau]
        s += np.sum(x1 * x2) / dN
    return s / (dT - tau)


def acl_spectrum_direct(X, scale):
    """l2hmc/utils/func_utils.py:114-116: [autocovariance(X / scale, tau) for tau in range(n - 1)]."""
    X = np.asarray(X, dtype=np.float64) / scale
    return np.array([autocovariance_direct(X, t) for t in range(X.shape[0] - 1)])


def autocorr_direct(x):
    """l2hmc/utils/autocorr.py:36-40: np.correlate(x, x, 'full') / max, non-negative lags (no mean removal)."""
    x = np.asarray(x, dtype=np.float64)
    n = len(x)
    full = np.array([np.dot(x[: n - k], x[k:]) for k in range(n)])
    return full / full.max()
