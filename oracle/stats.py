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
