"""Knee-point pick for the regularisation sweep (host side, 99 numbers).

The reference delegates this to ``kneed.KneeLocator(x, y, direction='increasing').knee``
(optimal_interpolation.py:37-39; ``kneed==0.8.3`` per requirements.txt:9, a third-party package
that is not part of the reference tree).  This is a dependency-free restatement of that
package's published "Kneedle" procedure for its default arguments (S=1, curve='concave',
interp_method='interp1d', online=False).  PARITY UNPINNED: the package cannot be installed in
the build environment, so the chosen index is not verified against it -- ``OI(...)`` therefore
also accepts ``reg_index=`` to bypass the pick, and always reports the index it used.
"""
from __future__ import annotations

import numpy as np


def _piecewise_linear_at_nodes(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """scipy ``interp1d(x, y)(x)``: the segment *left* of each node evaluated at the node
    (slope*(x_i - x_{i-1}) + y_{i-1}), which can differ from y_i in the last bit."""
    out = y.copy()
    if x.size > 1:
        slope = (y[1:] - y[:-1]) / (x[1:] - x[:-1])
        out[1:] = slope * (x[1:] - x[:-1]) + y[:-1]
    return out


def _unit_range(a: np.ndarray) -> np.ndarray:
    lo, hi = np.min(a), np.max(a)
    return (a - lo) / (hi - lo)


def _relative_extrema(d: np.ndarray, greater: bool) -> np.ndarray:
    """``scipy.signal.argrelextrema(d, np.greater_equal | np.less_equal)`` with order=1 and
    mode='clip' (each end point is compared with itself on its open side)."""
    left = np.concatenate((d[:1], d[:-1]))
    right = np.concatenate((d[1:], d[-1:]))
    if greater:
        m = (d >= left) & (d >= right)
    else:
        m = (d <= left) & (d <= right)
    return np.flatnonzero(m)


def knee_index(x, y, S: float = 1.0):
    """Index of the knee of an increasing, concave curve, or ``None`` when Kneedle finds none."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.size < 3:
        return None
    with np.errstate(all="ignore"):
        xn = _unit_range(x)
        yn = _unit_range(_piecewise_linear_at_nodes(x, y))
        diff = yn - xn
        peaks = _relative_extrema(diff, True)
        dips = _relative_extrema(diff, False)
        cut = diff[peaks] - S * abs(np.mean(np.diff(xn)))
    if peaks.size == 0:
        return None
    is_peak = np.zeros(x.size, dtype=bool)
    is_peak[peaks] = True
    is_dip = np.zeros(x.size, dtype=bool)
    is_dip[dips] = True
    threshold = np.nan
    at = None
    seen = 0
    for i in range(int(peaks[0]), x.size):
        if xn[i] == 1.0:                       # walked off the end without dropping below a threshold
            return None
        if is_peak[i]:
            threshold = cut[seen]
            at = i
            seen += 1
        if is_dip[i]:
            threshold = 0.0
        if diff[i + 1] < threshold:
            return at
    return None
