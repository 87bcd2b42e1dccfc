"""``OI`` -- the element-wise optimal-interpolation analysis, on the MI355X.

Drop-in for ``oisatgmi/optimal_interpolation.py:6-52`` of the reference: same signature, same
4-tuple ``(Xb, averaging_kernel, increment, sqrt(Sb))``, same in-place clamp of ``Y`` (:14), same
NaN semantics (unobserved cells come back NaN, ``Sa*reg == 0`` gives AK = NaN), same two prints.
The arithmetic runs in ``oisat_oi_curve`` / ``oisat_oi_apply`` (csrc/oi_diag.hip); only the
99-number knee pick happens on the host (``_kneedle.knee_index``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _hip
from ._kneedle import knee_index

#: what the last ``OI`` call chose -- {'index', 'scale', 'curve', 'knee_found'}
last_regularization = {}


def scaling_factors(regularization_on=True) -> np.ndarray:
    """The sweep of prior-error scalings, optimal_interpolation.py:15-20."""
    if regularization_on == True:                       # noqa: E712  (the reference compares with ==)
        return np.arange(0.1, 10, 0.1)
    return np.array([1.0])


def _curve(ctx, code, dSa, dSo, n, factors):
    sc = np.ascontiguousarray(factors, dtype=np.float64)
    mean = np.empty(sc.size, dtype=np.float64)
    cnt = np.empty(sc.size, dtype=np.int64)
    ctx.check(ctx.lib.oisat_oi_curve(ctx.h, code, dSa, dSo, n, sc.ctypes.data_as(C.POINTER(C.c_double)), sc.size,
                                     mean.ctypes.data_as(C.POINTER(C.c_double)),
                                     cnt.ctypes.data_as(C.POINTER(C.c_int64))))
    return mean, cnt


def regularization_pick(Sa, So, reg_index=None, ctx=None):
    """The reference's regularisation choice by itself (optimal_interpolation.py:15-41): the 99-scaling sweep of the
    element-wise averaging-kernel mean on the device, then the knee pick (or ``reg_index``).  -> (index, scale, curve,
    knee_found).  Used by the dense / tiled analysis modes, which take their prior-error scaling from it."""
    ctx = ctx or _hip.context()
    dt = _hip.compute_dtype(Sa, So)
    n = int(np.size(Sa))
    buf = ctx.alloc(2 * n * dt.itemsize)
    ctx.upload_into(buf.at(0), np.ravel(Sa), dtype=dt)
    ctx.upload_into(buf.at(n * dt.itemsize), np.ravel(So), dtype=dt)
    factors = scaling_factors(True)
    curve, _ = _curve(ctx, _hip.dtype_code(dt), buf.at(0), buf.at(n * dt.itemsize), n, factors)
    buf.free()
    found = False
    if reg_index is None:
        k = knee_index(factors, curve)
        found = k is not None
        index = 0 if k is None else int(k)
    else:
        index = int(reg_index)
    return index, float(factors[index]), curve, found


class DiagOI:
    """Device-resident element-wise OI: fields stay in HBM between calls (what ``OI`` does per call,
    minus the PCIe copies).  ``load`` once, ``run`` many times; ``download`` when needed."""

    def __init__(self, n: int, dtype=np.float32, ctx=None):
        self.ctx = ctx or _hip.context()
        self.dt = np.dtype(dtype)
        self.code = _hip.dtype_code(self.dt)
        self.n = int(n)
        item = self.dt.itemsize
        self.pool = self.ctx.alloc(8 * self.n * item)     # Xa | Y | Sa | So | Xb | AK | inc | err
        self.p = [self.pool.at(i * self.n * item) for i in range(8)]

    def load(self, Xa, Y, Sa, So):
        for dst, a in zip(self.p[:4], (Xa, Y, Sa, So)):
            self.ctx.upload_into(dst, np.ravel(a), dtype=self.dt)

    def run(self, regularization_on=True, reg_index=None):
        ctx = self.ctx
        factors = scaling_factors(regularization_on)
        curve = None
        index = 0
        if regularization_on == True:                     # noqa: E712
            curve, _ = _curve(ctx, self.code, self.p[2], self.p[3], self.n, factors)
            if reg_index is None:
                k = knee_index(factors, curve)
                index = 0 if k is None else int(k)
            else:
                index = int(reg_index)
        p = self.p
        ctx.check(ctx.lib.oisat_oi_apply(ctx.h, self.code, p[0], p[1], p[2], p[3], self.n, float(factors[index]),
                                         p[4], p[5], p[6], p[7]))
        return index, curve

    def run_fused(self, regularization_on=True, reg_index=None):
        """Sweep + knee pick + analysis entirely on the device (``oisat_oi_fused``): no host round
        trip, nothing to wait for.  ``fused_result()`` reads back the chosen index and the curve."""
        ctx = self.ctx
        factors = np.ascontiguousarray(scaling_factors(regularization_on), dtype=np.float64)
        if not hasattr(self, "_aux"):
            self._aux = ctx.alloc(8 * _hip.MAX_SCALES + 16)          # curve | index
        p = self.p
        forced = -1 if reg_index is None else int(reg_index)
        if regularization_on != True:                                 # noqa: E712
            forced = 0
        ctx.check(ctx.lib.oisat_oi_fused(ctx.h, self.code, p[0], p[1], p[2], p[3], self.n,
                                         factors.ctypes.data_as(C.POINTER(C.c_double)), factors.size, forced,
                                         p[4], p[5], p[6], p[7], self._aux.at(8 * _hip.MAX_SCALES), self._aux.ptr))
        self._nf = factors.size

    def fused_result(self):
        curve = self.ctx.download(self._aux.ptr, (self._nf,), np.float64)
        idx = int(self.ctx.download(self._aux.at(8 * _hip.MAX_SCALES), (1,), np.int32)[0])
        return idx, curve

    def download(self, shape):
        out = self.ctx.download(self.p[4], (4,) + tuple(shape), self.dt)
        return out[0], out[1], out[2], out[3]

    @staticmethod
    def algorithmic_bytes(n: int, itemsize: int) -> int:
        """read Xa, Y, Sa, So + write Xb, AK, inc, err (32 B/cell in fp32);
        the sweep re-reads Sa and So once more."""
        return 8 * n * itemsize


def OI(Xa: np.ndarray, Y: np.ndarray, Sa: np.ndarray, So: np.ndarray, regularization_on=True, reg_index=None):
    '''
    Optimal interpolation between two variables looking at the exact quantity (K = ones):
            Xb = Xa + Sa K^T (K Sa K^T + So)^-1 (Y - K Xa)

    ``reg_index`` (extension): force the index into the scaling sweep instead of the knee pick.
    '''
    print('Optimal interpolation begins...')
    Y[Y < 0] = 0.0                                       # in place on the caller's array, like the reference
    ctx = _hip.context()
    dt = _hip.compute_dtype(Xa, Y, Sa, So)
    code = _hip.dtype_code(dt)
    shape = np.shape(Xa)
    n = int(np.size(Xa))
    if not (np.shape(Y) == shape and np.shape(Sa) == shape and np.shape(So) == shape):
        # the reference would broadcast; the fused kernels want one shape
        Xa, Y2, Sa, So = np.broadcast_arrays(Xa, Y, Sa, So)
        shape = Xa.shape
        n = Xa.size
    else:
        Y2 = Y
    item = dt.itemsize
    pool = ctx.alloc(8 * n * item)                        # Xa | Y | Sa | So | Xb | AK | inc | err
    p = [pool.at(i * n * item) for i in range(8)]
    for dst, a in zip(p[:4], (Xa, Y2, Sa, So)):
        ctx.upload_into(dst, a, dtype=dt)

    factors = scaling_factors(regularization_on)
    found = False
    curve = None
    if regularization_on == True:                         # noqa: E712
        curve, _ = _curve(ctx, code, p[2], p[3], n, factors)
        if reg_index is None:
            k = knee_index(factors, curve)
            found = k is not None
            index = 0 if k is None else int(k)            # empty match -> [0], optimal_interpolation.py:40-41
        else:
            index = int(reg_index)
    else:
        index = 0
    scale = float(factors[index])
    print("The regularization factor is " + str(factors[index]))
    ctx.check(ctx.lib.oisat_oi_apply(ctx.h, code, p[0], p[1], p[2], p[3], n, scale, p[4], p[5], p[6], p[7]))
    out = ctx.download(p[4], (4,) + tuple(shape), dt)
    pool.free()
    last_regularization.clear()
    last_regularization.update(index=index, scale=scale, curve=curve, knee_found=found)
    return out[0], out[1], out[2], out[3]
