"""Chain statistics with the reference's definitions (host post-processing, O(chain length)):
  l2hmc/utils/func_utils.py:45-54   autocovariance (uncentred, normalised by the number of chains)
  l2hmc/utils/func_utils.py:114-116 acl_spectrum
  l2hmc/utils/func_utils.py:118-120 ESS
BASELINE.json's secondary metric (ESS/sec) is reported with these estimators."""
import numpy as np


def autocovariance(X, tau=0):
    """X: [steps, chains, dims] -> mean_t( sum_{chains,dims} x_t * x_{t+tau} / n_chains )."""
    X = np.asarray(X, dtype=np.float64)
    dT, dN, _ = X.shape
    if tau >= dT:
        raise ValueError("tau must be smaller than the number of steps")
    return float(np.sum(X[:dT - tau] * X[tau:]) / dN / (dT - tau))


def acl_spectrum(X, scale):
    """autocovariance(X / scale, tau) for tau = 0 .. n-2."""
    X = np.asarray(X, dtype=np.float64) / scale
    return np.array([autocovariance(X, tau=t) for t in range(X.shape[0] - 1)])


def ESS(A):
    """1 / (1 + 2 sum_{tau>=1} A_tau [A_tau > 0.05])."""
    A = np.asarray(A, dtype=np.float64)
    A = A * (A > 0.05)
    return 1. / (1. + 2 * np.sum(A[1:]))
