"""Chain statistics with the reference's definitions (host post-processing, O(chain length)):
  l2hmc/utils/func_utils.py:45-54   autocovariance (uncentred, normalised by the number of chains)
  l2hmc/utils/func_utils.py:114-116 acl_spectrum
  l2hmc/utils/func_utils.py:118-120 ESS
BASELINE.json's secondary metric (ESS/sec) is reported with these estimators."""
import numpy as np


def _is_tensor(x):
    try:
        import torch
        return isinstance(x, torch.Tensor)
    except ImportError:      # pragma: no cover
        return False


def autocovariance(X, tau=0):
    """X: [steps, chains, dims] -> mean_t( sum_{chains,dims} x_t * x_{t+tau} / n_chains ).
    A (device) tensor history is reduced where it lives; the result is a Python float either way."""
    if _is_tensor(X):
        import torch
        X = X.to(torch.float64)
        dT, dN = X.shape[0], X.shape[1]
        if tau >= dT:
            raise ValueError("tau must be smaller than the number of steps")
        return float(torch.sum(X[:dT - tau] * X[tau:]) / dN / (dT - tau))
    X = np.asarray(X, dtype=np.float64)
    dT, dN, _ = X.shape
    if tau >= dT:
        raise ValueError("tau must be smaller than the number of steps")
    return float(np.sum(X[:dT - tau] * X[tau:]) / dN / (dT - tau))


def acl_spectrum(X, scale):
    """autocovariance(X / scale, tau) for tau = 0 .. n-2.  A tensor history (e.g. the samples a device-resident
    run kept in HBM) goes through ONE batched FFT on its device -- all lags of all chains and dimensions at once,
    O(n log n) instead of the reference's O(n^2) Python loop -- and only the n-1 numbers come back."""
    if _is_tensor(X):
        import torch
        X = X.to(torch.float64) / scale
        n, dN = X.shape[0], X.shape[1]
        flat = X.reshape(n, -1)
        f = torch.fft.rfft(flat, n=2 * n, dim=0)
        acf = torch.fft.irfft(f * f.conj(), n=2 * n, dim=0)[:n].sum(dim=1)          # sum_t x_t x_{t+tau}, summed over series
        lags = torch.arange(n, device=X.device, dtype=torch.float64)
        return (acf / dN / (n - lags))[: n - 1].cpu().numpy()
    X = np.asarray(X, dtype=np.float64) / scale
    return np.array([autocovariance(X, tau=t) for t in range(X.shape[0] - 1)])


def ESS(A):
    """1 / (1 + 2 sum_{tau>=1} A_tau [A_tau > 0.05])."""
    A = np.asarray(A, dtype=np.float64)
    A = A * (A > 0.05)
    return 1. / (1. + 2 * np.sum(A[1:]))


# ---------------------------------------------------------------------------------------------
# Integrated autocorrelation time (l2hmc/utils/autocorr.py:23-199), batched over chains.
# The reference loops over walkers and parameters in Python with one FFT each; here every series
# of a [steps, chains, dims] history goes through ONE batched FFT (NumPy on the host, or torch.fft
# on the device the history already lives on when a tensor is passed).
# ---------------------------------------------------------------------------------------------
class AutocorrError(Exception):
    """Chain too short for a reliable estimate (autocorr.py:181-196); carries the estimate."""

    def __init__(self, tau, *args):
        self.tau = tau
        super().__init__(*args)


def next_pow_two(n):
    """Smallest power of two >= n (autocorr.py:82-87)."""
    return 1 << max(0, int(n) - 1).bit_length()


def _acf_batched(x, nfft):
    """Unnormalised circular-padded autocorrelation along axis 0 of x [n, ...] (mean removed)."""
    try:
        import torch
        if isinstance(x, torch.Tensor):
            x = x.to(torch.float64)
            x = x - x.mean(dim=0, keepdim=True)
            f = torch.fft.rfft(x, n=nfft, dim=0)
            return torch.fft.irfft(f * f.conj(), n=nfft, dim=0)[: x.shape[0]].cpu().numpy()
    except ImportError:      # pragma: no cover
        pass
    x = np.asarray(x, dtype=np.float64)
    x = x - x.mean(axis=0, keepdims=True)
    f = np.fft.rfft(x, n=nfft, axis=0)
    return np.fft.irfft(f * np.conj(f), n=nfft, axis=0)[: x.shape[0]]


def autocorr_func_1d(x):
    """Normalised autocorrelation function of series along axis 0 (autocorr.py:107-126; a 1-D input
    gives the reference's result, more axes are treated as independent series)."""
    if not hasattr(x, "shape") or len(x.shape) == 0:
        x = np.atleast_1d(x)
    acf = _acf_batched(x, 2 * next_pow_two(x.shape[0]))
    with np.errstate(invalid='ignore', divide='ignore'):      # a constant series gives 0 / 0 = nan, as in the reference
        return acf / acf[0]


def autocorr_fast(X, kappa=500):
    """autocorr.py:23-34: FFT autocorrelation with the unbiased 1/(N-k) weights, truncated at kappa."""
    if not _is_tensor(X):
        X = np.asarray(X, dtype=np.float64)
    N = X.shape[0]
    acf = _acf_batched(X, 2 * N)
    acf = acf / (N - np.arange(N)).reshape((-1,) + (1,) * (acf.ndim - 1))
    return (acf / acf[0])[:kappa]


def autocorr(X):
    """autocorr.py:36-40: np.correlate(X, X, 'full') normalised by its maximum, non-negative lags
    (no mean removal, as the reference)."""
    if _is_tensor(X):
        import torch
        X = X.to(torch.float64)
        n = X.shape[0]
        f = torch.fft.rfft(X, n=2 * n, dim=0)
        full = torch.fft.irfft(f * f.conj(), n=2 * n, dim=0)[:n].cpu().numpy()
        return full / full[0]
    X = np.asarray(X, dtype=np.float64)
    n = X.shape[0]
    f = np.fft.rfft(X, n=2 * n)
    full = np.fft.irfft(f * np.conj(f), n=2 * n)[:n]
    return full / full[0]                # the maximum of an autocorrelation sits at lag 0


def calc_iat(X, kappa=500):
    """autocorr.py:70-76 -> (tau, curve) with tau = 1 + 2 sum(curve) (the lag-0 term included, as written)."""
    curve = autocorr_fast(X, kappa)
    return 1 + 2 * np.sum(curve, axis=0), curve


def auto_window(taus, c):
    """autocorr.py:128-132: first lag m with m >= c * tau(m)."""
    m = np.arange(len(taus)) < c * taus
    return int(np.argmin(m)) if np.any(m) else len(taus) - 1


def integrated_time(x, c=5, tol=50, quiet=False):
    """autocorr.py:134-199 (Sokal's automatic windowing, averaged over walkers): x [steps] |
    [steps, walkers] | [steps, walkers, dims] -> (tau per dim, flag)."""
    if not hasattr(x, "shape"):
        x = np.atleast_1d(x)
    if len(x.shape) == 1:
        x = x[:, None, None]
    if len(x.shape) == 2:
        x = x[:, :, None]
    if len(x.shape) != 3:
        raise ValueError("invalid dimensions")
    n_t, _, n_d = x.shape
    f = autocorr_func_1d(x).mean(axis=1)                    # [steps, dims]
    taus = 2.0 * np.cumsum(f, axis=0) - 1.0
    windows = np.array([auto_window(taus[:, d], c) for d in range(n_d)])
    tau_est = taus[windows, np.arange(n_d)]
    flag = None
    if np.any(tol * tau_est > n_t):
        msg = (f"The chain is shorter than {tol} times the integrated autocorrelation time for "
               f"{int(np.sum(tol * tau_est > n_t))} parameter(s). N/{tol} = {n_t / tol:.0f}; tau: {tau_est}")
        if not quiet:
            raise AutocorrError(tau_est, msg)
        flag = 1
    return tau_est, flag


def autocorr_gw2010(y, c=5.0):
    """autocorr.py:89-94 (Goodman & Weare 2010): y [walkers, steps]."""
    f = autocorr_func_1d(np.mean(np.asarray(y, dtype=np.float64), axis=0))
    taus = 2. * np.cumsum(f) - 1.0
    return taus[auto_window(taus, c)]


def autocorr_new(y, c=5.0):
    """autocorr.py:96-103: y [walkers, steps], walker-averaged autocorrelation function."""
    f = autocorr_func_1d(np.asarray(y, dtype=np.float64).T).mean(axis=1)
    taus = 2. * np.cumsum(f) - 1.
    return taus[auto_window(taus, c)]


# ---------------------------------------------------------------------------------------------
# Run statistics (l2hmc/gauge_model.py:1473-1531 `calc_observables_stats`) and block-jackknife errors
# (l2hmc/utils/data_utils.py:66-143); plain NumPy, no scikit-learn / scipy dependency.
# ---------------------------------------------------------------------------------------------
def sem(a):
    """scipy.stats.sem along axis 0: sample standard deviation (ddof = 1) / sqrt(n)."""
    a = np.asarray(a, dtype=np.float64)
    return a.std(axis=0, ddof=1) / np.sqrt(a.shape[0])


def block_resampling(data, num_blocks):
    """data_utils.py:66-84: the `num_blocks` leave-one-block-out resamples of `data` along axis 0.  Blocks are the
    contiguous folds of sklearn.model_selection.KFold without shuffling: the first n % k folds hold n // k + 1
    samples, the rest n // k."""
    data = np.asarray(data)
    n = data.shape[0]
    if n < 1:
        raise ValueError("Data must have at least one sample.")
    if num_blocks < 1:
        raise ValueError("Number of resampled blocks must be greater than or equal to 1.")
    if n < num_blocks:
        num_blocks = max(2, n)
    if num_blocks > n:
        raise ValueError(f"Cannot have number of splits n_splits={num_blocks} greater than the number of samples: {n}.")
    sizes = np.full(num_blocks, n // num_blocks)
    sizes[: n % num_blocks] += 1
    out, start = [], 0
    for sz in sizes:
        out.append(np.concatenate([data[:start], data[start + sz:]]))
        start += sz
    return out


def jackknife_err(y_i, y_full, num_blocks):
    """data_utils.py:106-116: sqrt( sum (y_i - y_full)^2 / (num_blocks - 1) * num_blocks )  (as written)."""
    y_i, y_full = np.asarray(y_i), np.asarray(y_full)
    return np.sqrt(np.sum((y_i - y_full) ** 2) / (num_blocks - 1) * num_blocks)


def calc_avg_vals_errors(data, num_blocks=100):
    """data_utils.py:119-143 -> (mean, block-jackknife error)."""
    arr = np.asarray(data)
    avg = np.mean(arr)
    rs = [np.mean(b) for b in block_resampling(arr, num_blocks)]
    return avg, jackknife_err(rs, avg, num_blocks)


def calc_observables_stats(actions, plaqs, charges, therm_frac=10):
    """gauge_model.py:1473-1531 on [steps, chains] histories (what GaugeSampler.run returns): drop the first
    steps // therm_frac steps, per-chain means and standard errors of action, plaquette, topological charge and
    susceptibility Q^2, and the charge histogram.  As in the reference the susceptibility series is NOT
    thermalisation-trimmed (it is squared before the cut)."""
    actions, plaqs = np.asarray(actions, dtype=np.float64), np.asarray(plaqs, dtype=np.float64)
    charges = np.asarray(np.asarray(charges), dtype=int)           # np.array(..., dtype=int) truncates like the reference
    suscept = charges ** 2
    therm = actions.shape[0] // therm_frac
    actions, plaqs, charges = actions[therm:], plaqs[therm:], charges[therm:]
    vals, counts = np.unique(charges, return_counts=True)
    probs = {int(v): float(c) / counts.sum() for v, c in zip(vals, counts)}
    return ((actions.mean(axis=0), sem(actions)), (plaqs.mean(axis=0), sem(plaqs)),
            (charges.mean(axis=0), sem(charges)), (suscept.mean(axis=0), sem(suscept)), probs)
