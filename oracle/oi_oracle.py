"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  NOT PART OF THE PRODUCT PATH.

float64 NumPy/SciPy restatement of the reference's optimal-interpolation hot path
(ahsouri/OI-SAT-GMI; citations below are ``file:line`` into the reference tree).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; nothing under ``oi-sat-gmi_amd/`` does, and the product fails loudly when its HIP
library is missing instead of falling back to anything in here.

Pinning status ("Oracle"):

* ``OI`` (fixed index), the 99-point regularisation curve, ``error_averager``, ``averaging``,
  ``_upscaler``, ``_interpolosis`` types 1-4 and ``interpolator`` (2-D field path) are PINNED:
  ``tests/golden/*.npz`` hold outputs of the reference's own functions, produced in the build
  container by ``tests/golden/make_golden.py`` importing the reference modules.
* ``amf_recal`` and ``ak_conv`` (MOPITT / GOSAT averaging-kernel convolution) are PINNED the same way
  (``amf_recal.npz``, ``ak_conv.npz``), and so is ``pwv_calculator`` (``pwv.npz``).
* The knee index chosen from that curve comes from the third-party package ``kneed==0.8.3``
  (requirements.txt:9; call site optimal_interpolation.py:37-39), which is neither vendored in the
  reference nor installed here: ``kneedle_knee`` restates its published algorithm and is
  **parity unpinned**.
* ``dense_oi`` (Gaussian-B, K = B H^T (H B H^T + R)^-1) has no counterpart in the reference at
  all (its OI is element-wise, optimal_interpolation.py:8-10,27): **parity unpinned**; its only
  reference-anchored check is the L -> 0 / H = selection limit against ``OI``.
"""
from __future__ import annotations

import datetime

import numpy as np
from scipy import linalg as _sla
from scipy.interpolate import interp1d as _interp1d
from scipy.signal import argrelextrema as _argrelextrema
from scipy.spatial import cKDTree as _cKDTree

EARTH_RADIUS_KM = 6371.0


# --------------------------------------------------------------------------------------------
# Kneedle (kneed==0.8.3 KneeLocator, defaults S=1.0, curve='concave', interp_method='interp1d',
# online=False, with direction='increasing' as passed at optimal_interpolation.py:37-38)
# --------------------------------------------------------------------------------------------
def kneedle_knee(x, y, S: float = 1.0):
    """Return ``(knee_x or None, index or None)``.  PARITY UNPINNED (kneed not available)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    with np.errstate(all="ignore"):
        ds_y = _interp1d(x, y)(x)                              # step 1: "smooth" line through the data
        xn = (x - x.min()) / (x.max() - x.min())               # step 2: normalise
        yn = (ds_y - ds_y.min()) / (ds_y.max() - ds_y.min())
        yd = yn - xn                                           # step 3: difference curve (concave/increasing: no flip)
        maxima = _argrelextrema(yd, np.greater_equal)[0]       # step 4
        minima = _argrelextrema(yd, np.less_equal)[0]
        tmx = yd[maxima] - S * np.abs(np.diff(xn).mean())      # step 5
    if maxima.size == 0:
        return None, None
    thr = None
    thr_idx = None
    mi = 0
    for i in range(xn.size):                                   # step 6
        if i < maxima[0]:
            continue
        if xn[i] == 1.0:
            break
        if (maxima == i).any():
            thr = tmx[mi]
            thr_idx = i
            mi += 1
        if (minima == i).any():
            thr = 0.0
        if yd[i + 1] < thr:
            return float(x[thr_idx]), int(thr_idx)
    return None, None


def scaling_factors(regularization_on=True):
    """optimal_interpolation.py:15-20."""
    if regularization_on:
        return list(np.arange(0.1, 10, 0.1))
    return [1.0]


# --------------------------------------------------------------------------------------------
# OI  (optimal_interpolation.py:6-52)
# --------------------------------------------------------------------------------------------
def oi_curve(Sa, So, factors):
    """The per-scaling mean averaging kernel, optimal_interpolation.py:26-33 (values only)."""
    means = []
    with np.errstate(all="ignore"):
        for reg in factors:
            t = Sa * float(reg)
            k = t * (t + So) ** (-1)
            sb = (np.ones_like(k) - k) * t
            ak = np.ones_like(sb) - sb / t
            means.append(np.nanmean(ak.flatten()))
    return np.array(means)


def oi_fields(Xa, Y, Sa, So, reg):
    """K, AK, Sb for one scaling and the analysis that follows, :27-31 and :49-52."""
    with np.errstate(all="ignore"):
        t = Sa * float(reg)
        k = t * (t + So) ** (-1)
        sb = (np.ones_like(k) - k) * t
        ak = np.ones_like(sb) - sb / t
        inc = k * (Y - Xa)
        xb = Xa + inc
        return xb, ak, inc, np.sqrt(sb)


def OI(Xa, Y, Sa, So, regularization_on=True, forced_index=None):
    """Restatement of ``OI``.  Mutates ``Y`` in place like the reference (:14).

    Returns ``(Xb, AK, increment, sqrt(Sb), curve, index)``; the first four are the reference's
    return tuple.  ``forced_index`` bypasses the (unpinned) knee pick.
    """
    Y[Y < 0] = 0.0
    factors = scaling_factors(regularization_on)
    curve = oi_curve(Sa, So, factors)
    if forced_index is not None:
        index = int(forced_index)
    elif regularization_on:
        _, idx = kneedle_knee(np.array(factors), curve)
        index = 0 if idx is None else idx                      # :39-41 empty match -> [0]
    else:
        index = 0
    xb, ak, inc, err = oi_fields(Xa, Y, Sa, So, factors[index])
    return xb, ak, inc, err, curve, index


# --------------------------------------------------------------------------------------------
# averaging.py
# --------------------------------------------------------------------------------------------
def error_averager(error_X):
    """averaging.py:11-24 vectorised: per cell inf->NaN, drop NaN, sqrt(sum/count^2)."""
    e = np.array(error_X, dtype=np.float64, copy=True)
    e[np.isinf(e)] = np.nan
    valid = ~np.isnan(e)
    cnt = valid.sum(axis=0).astype(np.float64)
    s = np.where(valid, e, 0.0).sum(axis=0)
    with np.errstate(all="ignore"):
        return np.sqrt(s / cnt ** 2)


def averaging(startdate: str, enddate: str, reader_obj, amf_type=None, opt_type=None):
    """averaging.py:26-120 for the single-(month, year) windows job.py:77-82 passes, with the
    reference's quirks: ``None`` granules skipped (:50-51,:73-74); the sat-vcd accumulator starts
    from zeros, the others from NaN (:53-63); inf->NaN on sat vcd only (:92); the reduction block
    sits at year-loop level and therefore sees the LAST month's lists (:97-108).
    ``amf_type`` / ``opt_type``: the record classes to dispatch aux1/aux2 on (:82-90)."""
    sd = datetime.date(int(startdate[0:4]), int(startdate[5:7]), int(startdate[8:10]))
    ed = datetime.date(int(enddate[0:4]), int(enddate[5:7]), int(enddate[8:10]))
    days = [sd + datetime.timedelta(n) for n in range(int((ed - sd).days))]
    months = np.array([d.month for d in days])
    years = np.array([d.year for d in days])
    first = next(g for g in reader_obj.sat_data if g is not None)
    ny, nx = np.shape(first.latitude_center)[0:2]
    nm = months.max() - months.min() + 1
    nyr = years.max() - years.min() + 1
    sat_vcd = np.zeros((ny, nx, nm, nyr))
    sat_err = np.full_like(sat_vcd, np.nan)
    ctm_vcd = np.full_like(sat_vcd, np.nan)
    aux1 = np.full_like(sat_vcd, np.nan)
    aux2 = np.full_like(sat_vcd, np.nan)
    times = []
    for year in range(years.min(), years.max() + 1):
        for month in range(months.min(), months.max() + 1):
            v, e, c, a1, a2, times = [], [], [], [], [], []
            for g in reader_obj.sat_data:
                if g is None:
                    continue
                if g.time.year == year and g.time.month == month:
                    times.append(g.time)
                    v.append(g.vcd)
                    e.append(g.uncertainty)
                    c.append(g.ctm_vcd)
                    if amf_type is not None and isinstance(g, amf_type):
                        a1.append(g.new_amf)
                        a2.append(g.old_amf)
                    elif opt_type is not None and isinstance(g, opt_type):
                        a1.append(g.x_col)
                        a2.append(g.ctm_xcol)
                    else:
                        a1.append(np.nan * g.vcd)
                        a2.append(np.nan * g.vcd)
            v = np.array(v, dtype=np.float64)
            v[np.isinf(v)] = np.nan
            e = np.array(e, dtype=np.float64)
            c = np.array(c, dtype=np.float64)
            a1 = np.array(a1, dtype=np.float64)
            a2 = np.array(a2, dtype=np.float64)
        mi = month - months.min()
        yi = year - years.min()
        with np.errstate(all="ignore"):
            if v.size != 0:
                sat_vcd[:, :, mi, yi] = np.nanmean(v, axis=0)
                sat_err[:, :, mi, yi] = error_averager(e ** 2)
                ctm_vcd[:, :, mi, yi] = np.nanmean(c, axis=0)
            if a1.size != 0:
                aux1[:, :, mi, yi] = np.nanmean(a1, axis=0)
                aux2[:, :, mi, yi] = np.nanmean(a2, axis=0)
    ts = [t.timestamp() for t in times]
    avg_dt = datetime.datetime.fromtimestamp(sum(ts) / len(ts))
    return (sat_vcd.squeeze(), sat_err.squeeze(), ctm_vcd.squeeze(), aux1.squeeze(),
            aux2.squeeze(), avg_dt)


# --------------------------------------------------------------------------------------------
# driver.py hot-path wrappers
# --------------------------------------------------------------------------------------------
BIAS_TABLE = {  # driver.py:68-100  (offset, slope)
    ("TROPOMI", "NO2"): (0.32, 0.66),
    ("TROPOMI", "HCHO"): (0.90, 0.59),
    ("OMI", "NO2"): (0.32, 0.63),
    ("OMI", "HCHO"): (0.821, 0.79),
}


def bias_correct(sat_vcd, sat_type, gasname):
    """driver.py:65-106."""
    if (sat_type, gasname) in BIAS_TABLE:
        off, slope = BIAS_TABLE[(sat_type, gasname)]
        return (sat_vcd - off) / slope
    return sat_vcd


def driver_oi_inputs(ctm_vcd, sat_vcd, sat_err, aux1, aux2, sensor, error_ctm=50.0):
    """Argument wiring of ``oisatgmi.oi`` (driver.py:108-114): returns (Xa, Y, Sa, So)."""
    if sensor != "GOSAT":
        return ctm_vcd, sat_vcd, (ctm_vcd * error_ctm / 100.0) ** 2, sat_err ** 2
    return aux2, aux1, (aux2 * error_ctm / 100.0) ** 2, sat_err ** 2


# --------------------------------------------------------------------------------------------
# interpolator.py
# --------------------------------------------------------------------------------------------
def boxfilter_symm(Z, ky: int, kx: int, variance: bool = False):
    """``signal.convolve2d(Z, ones(ky,kx)/(kx*ky)[**2], boundary='symm', mode='same')``
    (interpolator.py:40-46, :72-76): window [i - K//2, i + (K-1)//2] with edge-repeating
    reflection; NaN anywhere in the window poisons the output."""
    Z = np.asarray(Z, dtype=np.float64)
    Ny, Nx = Z.shape
    w = 1.0 / (kx * ky) ** (2 if variance else 1)
    iy = np.arange(-(ky // 2), Ny + (ky - 1) // 2)
    ix = np.arange(-(kx // 2), Nx + (kx - 1) // 2)

    def refl(i, n):
        i = np.where(i < 0, -i - 1, i)
        i = np.where(i >= n, 2 * n - 1 - i, i)
        return i

    P = Z[np.ix_(refl(iy, Ny), refl(ix, Nx))]
    out = np.zeros((Ny, Nx))
    for a in range(ky):
        for b in range(kx):
            out += P[a:a + Ny, b:b + Nx] * w
    return out


def interpolosis_nn(tree: _cKDTree, Z, X, Y, dists, threshold):
    """``_interpolosis`` types 2 and 4 (interpolator.py:17-20, :28-33): nearest-neighbour gather
    from the k-d tree's points, then NaN where ``dists > 2*threshold``."""
    tp = np.column_stack((X.ravel(), Y.ravel()))
    _, idx = tree.query(tp)
    ZZ = np.asarray(Z, dtype=np.float64).ravel()[idx].reshape(X.shape)
    ZZ[dists > threshold * 2.0] = np.nan
    return ZZ


def _tps(r):
    """thin-plate spline r^2 log r (0 at r = 0), scipy's default RBFInterpolator kernel"""
    with np.errstate(all="ignore"):
        return np.where(r == 0.0, 0.0, r * r * np.log(r))


def interpolosis_rbf(pts, Z, X, Y, dists, threshold, neighbors=5, only_unmasked=False):
    """``_interpolosis`` type 3 (interpolator.py:21-27): ``RBFInterpolator(points, Z.flatten(),
    neighbors=5)`` -- thin-plate-spline kernel, degree-1 polynomial tail, no smoothing (scipy
    1.11/1.15 ``_rbfinterp.py``) -- evaluated at every target, then NaN where ``dists > 2*threshold``.

    Per target: the k nearest data points (indices sorted ascending), the (k+3)x(k+3) system
    ``[[K, P], [P^T, 0]] c = [d, 0]`` with ``K_ij = tps(|y_i - y_j|)`` and ``P = [1, xhat, yhat]`` on
    coordinates shifted/scaled to [-1, 1] over the neighbourhood, then ``out = [tps(|x - y_i|), 1, xhat,
    yhat] . c``.  scipy solves each distinct neighbourhood with LAPACK dgesv; this restatement solves
    all targets as one batch (``np.linalg.solve``), same arithmetic up to rounding.  A singular system
    raises ``LinAlgError`` like scipy.  ``only_unmasked`` restricts the work (and the singularity check)
    to targets that survive the mask -- what the HIP backend does."""
    pts = np.asarray(pts, dtype=np.float64)
    d = np.asarray(Z, dtype=np.float64).ravel()
    k = int(min(neighbors, pts.shape[0]))
    if k < 3:
        raise ValueError("At least 3 data points are required when `degree` is 1 and the number of dimensions is 2.")
    tgt = np.column_stack((np.ravel(X), np.ravel(Y))).astype(np.float64)
    out = np.full(tgt.shape[0], np.nan)
    keep = ~(np.ravel(dists) > threshold * 2.0) if only_unmasked else np.ones(tgt.shape[0], dtype=bool)
    if keep.any():
        x = tgt[keep]
        from scipy.spatial import KDTree                              # RBFInterpolator's own tree class (leafsize 10): among
        _, nb = KDTree(pts).query(x, k)                               # equidistant candidates it is this tree's pick that counts
        nb = np.sort(nb.reshape(x.shape[0], k), axis=1)
        y = pts[nb]                                                   # (T, k, 2)
        mins, maxs = y.min(axis=1), y.max(axis=1)
        shift, scale = (maxs + mins) / 2, (maxs - mins) / 2
        scale[scale == 0.0] = 1.0
        yhat = (y - shift[:, None, :]) / scale[:, None, :]
        xhat = (x - shift) / scale
        T = x.shape[0]
        A = np.zeros((T, k + 3, k + 3))
        A[:, :k, :k] = _tps(np.sqrt(((y[:, :, None, :] - y[:, None, :, :]) ** 2).sum(-1)))
        A[:, :k, k] = 1.0
        A[:, :k, k + 1:] = yhat
        A[:, k:, :k] = np.swapaxes(A[:, :k, k:], 1, 2)
        rhs = np.zeros((T, k + 3))
        rhs[:, :k] = d[nb]
        bad = ~np.isfinite(rhs).all(axis=1)                           # NaN data poison the whole neighbourhood
        rhs[bad] = 0.0
        coef = np.linalg.solve(A, rhs[:, :, None])[:, :, 0]           # raises LinAlgError("Singular matrix")
        vec = np.empty((T, k + 3))
        vec[:, :k] = _tps(np.sqrt(((x[:, None, :] - y) ** 2).sum(-1)))
        vec[:, k] = 1.0
        vec[:, k + 1:] = xhat
        o = (vec * coef).sum(axis=1)
        o[bad] = np.nan
        out[keep] = o
    ZZ = out.reshape(np.shape(X))
    ZZ[dists > threshold * 2.0] = np.nan
    return ZZ


def upscaler(X, Y, Z, ctm_models_coordinate, grid_size, threshold, error=False):
    """``_upscaler`` (interpolator.py:48-97)."""
    clat = ctm_models_coordinate["Latitude"]
    clon = ctm_models_coordinate["Longitude"]
    dlon = np.abs(clon[0, 0] - clon[0, 1])
    dlat = np.abs(clat[0, 0] - clat[1, 0])
    if (dlon >= grid_size) or (dlat >= grid_size):
        kx = np.floor(dlon / grid_size)
        ky = np.floor(dlat / grid_size)
        kx = 1 if kx == 0 else int(kx)
        ky = 1 if ky == 0 else int(ky)
        Zf = boxfilter_symm(Z, ky, kx, variance=error)
        pts = np.column_stack((X.ravel(), Y.ravel()))
        tree = _cKDTree(pts)
        dists, _ = tree.query(np.stack([clon, clat], axis=-1))
        Zc = interpolosis_nn(tree, Zf, clon, clat, dists, threshold)
        return clon, clat, Zc, False
    return X, Y, Z, True


def interpolator(interpolator_type, grid_size, sat_data, ctm_models_coordinate, flag_thresh=0.75,
                 record_type=None):
    """``interpolator`` (interpolator.py:100-291) for both record kinds, interpolator types 1-4: the 2-D fields
    (vcd, amf, tropopause if array, uncertainty; :162-188), the per-level scattering-weight / pressure loops of a
    ``satellite_amf`` record (:191-213) and the ``satellite_opt`` branch (:216-283: a-priori column, surface pressure,
    a-priori surface, x_col, averaging kernels -- nz+1 rows for MOPITT, nz for GOSAT --, pressure weights (GOSAT only),
    pressure_mid, a-priori profile).  A record with an ``x_col`` attribute is taken as ``satellite_opt``.
    Returns a ``record_type`` (positional, :284-290) or None."""
    if interpolator_type not in (1, 2, 3, 4):
        raise Exception("other type of interpolation methods has not been implemented yet")
    clat = ctm_models_coordinate["Latitude"]
    clon = ctm_models_coordinate["Longitude"]
    dlon = np.abs(clon[0, 0] - clon[0, 1])
    dlat = np.abs(clat[0, 0] - clat[1, 0])
    threshold_ctm = np.sqrt(dlon ** 2 + dlat ** 2)
    mask = np.multiply(sat_data.quality_flag > flag_thresh, 1.0).squeeze()
    mask[mask != 1.0] = np.nan
    pts = np.column_stack((np.ravel(sat_data.longitude_center), np.ravel(sat_data.latitude_center)))
    lon_grid = np.arange(clon.min(), clon.max() + grid_size, grid_size)
    lat_grid = np.arange(clat.min(), clat.max() + grid_size, grid_size)
    lons, lats = np.meshgrid(lon_grid, lat_grid)
    tree = _cKDTree(pts)
    dists, _ = tree.query(np.stack([lons, lats], axis=-1))

    tri = None
    if interpolator_type == 1:
        from scipy.spatial import Delaunay
        from scipy.interpolate import LinearNDInterpolator
        try:
            tri = Delaunay(pts)                                 # interpolator.py:151-155
        except Exception:
            return None

    def regrid(field, error=False):
        if tri is not None:                                     # _interpolosis type 1, interpolator.py:12-16
            zz = LinearNDInterpolator(tri, np.asarray(field, dtype=np.float64).flatten(), fill_value=np.nan)((lons, lats))
            zz[dists > grid_size * 2.0] = np.nan
        elif interpolator_type == 3:
            zz = interpolosis_rbf(pts, field, lons, lats, dists, grid_size)
        else:
            zz = interpolosis_nn(tree, field, lons, lats, dists, grid_size)
        return upscaler(lons, lats, zz, ctm_models_coordinate, grid_size, threshold_ctm, error=error)

    with np.errstate(all="ignore"):
        ux, uy, vcd, need = regrid(sat_data.vcd * mask)
        if np.isnan(np.nanmean(vcd.flatten())):
            return None
        is_opt = hasattr(sat_data, "x_col")
        if not is_opt:
            _, _, amf, _ = regrid(sat_data.amf * mask)                          # :169-173 (satellite_amf only)
        if np.size(sat_data.tropopause) != 1:
            _, _, trop, _ = regrid(sat_data.tropopause * mask)
        else:
            trop = np.empty((1))
        _, _, unc, _ = regrid(sat_data.uncertainty ** 2 * mask, error=True)
        unc = np.sqrt(unc)
        if is_opt:                                                              # interpolator.py:216-287
            def levels(cube, count):
                return np.stack([regrid(np.squeeze(np.asarray(cube)[z]) * mask)[2] for z in range(count)])
            nz = np.shape(sat_data.pressure_mid)[0]
            single = {}
            for nm in ("aprior_column", "surface_pressure", "apriori_surface"):  # each only `if field.any()` (:219-235)
                if getattr(sat_data, nm).any():
                    single[nm] = regrid(getattr(sat_data, nm) * mask)[2]
            x_col = regrid(sat_data.x_col * mask)[2]                             # :237-240
            if sat_data.sensor == 'MOPITT':                                      # :241-250
                aks = levels(sat_data.averaging_kernels, nz + 1)
                pw = np.empty((1))
            if sat_data.sensor == 'GOSAT':                                       # :251-268
                aks = levels(sat_data.averaging_kernels, nz)
                pw = levels(sat_data.pressure_weight, nz)
            pm = levels(sat_data.pressure_mid, nz)                               # :270-277
            apro = levels(sat_data.apriori_profile, nz)                          # :278-283
            fields = (vcd, sat_data.time, [], trop, uy, ux, [], [], unc, [], pm, aks, need, [], [], [],
                      single["aprior_column"], apro, single["surface_pressure"], single["apriori_surface"], x_col, pw,
                      sat_data.sensor)                                           # :285-287
            return record_type(*fields) if record_type is not None else fields
        # per-level cubes of the two-step retrievals, interpolator.py:191-213
        if np.size(sat_data.scattering_weights) != 1:
            nz = np.shape(sat_data.pressure_mid)[0]
            sw = np.stack([regrid(np.squeeze(sat_data.scattering_weights[z]) * mask)[2] for z in range(nz)])
            pm = np.stack([regrid(np.squeeze(sat_data.pressure_mid[z]) * mask)[2] for z in range(nz)])
        else:
            sw = np.empty((1))
            pm = np.zeros((np.shape(sat_data.pressure_mid)[0], np.shape(ux)[0], np.shape(ux)[1]))
    fields = (vcd, amf, sat_data.time, trop, uy, ux, [], [], unc, [], pm, sw,
              need, [], [], [], [])
    return record_type(*fields) if record_type is not None else fields


# --------------------------------------------------------------------------------------------
# amf_recal.py  (upstream of the averaging)
# --------------------------------------------------------------------------------------------
def partial_column(deltap, profile):
    """amf_recal.py:51-56."""
    return deltap * profile / 9.80665 / 28.97e-3 * 6.02214076e23 * 1e-4 * 1e-15 * 100.0 * 1e-9


def _flat_time(t):
    return t.year * 10000 + t.month * 100 + t.day + t.hour / 24.0 + t.minute / 60.0 / 24.0 + t.second / 3600.0 / 24.0


def _hour_time(t):
    return t.hour / 24.0 + t.minute / 60.0 / 24.0 + t.second / 3600.0 / 24.0


def amf_pixel(sat_p, sat_sw, ctm_p, ctm_pc, trop):
    """One pixel of _vertical_interp_and_amf (amf_recal.py:101-118): returns (new_amf, model_vcd)."""
    f = _interp1d(np.log(sat_p), sat_sw, fill_value="extrapolate")
    with np.errstate(all="ignore"):
        sw = f(np.log(ctm_p))
    sw[np.isinf(sw)] = 0.0
    pc = np.array(ctm_pc, copy=True)                # keeps the model's dtype: np.nansum(float32) accumulates in float32
    if trop is not None:
        m = ctm_p < trop
        sw[m] = np.nan
        pc[m] = np.nan
    scd = np.nansum(sw * pc)
    vcd = np.nansum(pc)
    return (scd / vcd if vcd != 0 else np.nan), vcd


def amf_recal(ctm_data, sat_data):
    """Restatement of ``amf_recal`` (amf_recal.py:121-185) for granules already on the model grid or
    needing the model upscaled (:154-158).  Mutates and returns ``sat_data`` like the reference."""
    tc, th = [], []
    for rec in ctm_data:
        tc += [_flat_time(t) for t in rec.time]
        th += [_hour_time(t) for t in rec.time]
    tc, th = np.array(tc), np.array(th)
    for L2 in sat_data:
        if L2 is None:
            continue
        if not ctm_data[0].averaged:
            ci = int(np.argmin(np.abs(_flat_time(L2.time) - tc)))
            day, hour = int(np.floor(ci / 8.0)), int(ci % 8)
        else:
            ci = int(np.argmin(np.abs(_hour_time(L2.time) - th)))
            day, hour = 0, ci
        pmid = ctm_data[day].pressure_mid[hour].squeeze()
        prof = ctm_data[day].gas_profile[hour].squeeze()
        delp = ctm_data[day].delta_p[hour].squeeze()
        pc = partial_column(delp, prof)
        if L2.ctm_upscaled_needed:
            coord = {"Longitude": L2.longitude_center, "Latitude": L2.latitude_center}
            thr = np.sqrt(np.abs(coord["Longitude"][0, 0] - coord["Longitude"][0, 1]) ** 2 +
                          np.abs(coord["Latitude"][0, 0] - coord["Latitude"][1, 0]) ** 2)
            clon, clat = ctm_data[0].longitude, ctm_data[0].latitude
            gs = np.sqrt(np.abs(clon[0, 0] - clon[0, 1]) ** 2 + np.abs(clat[0, 0] - clat[1, 0]) ** 2)
            pmid = np.stack([upscaler(clon, clat, pmid[z], coord, gs, thr)[2] for z in range(pmid.shape[0])])
            pc = np.stack([upscaler(clon, clat, pc[z], coord, gs, thr)[2] for z in range(pc.shape[0])])
        has_trop = np.size(L2.tropopause) != 1
        if np.size(L2.scattering_weights) == 1:
            pc = np.array(pc, copy=True)
            if has_trop:
                for z in range(pc.shape[0]):
                    pc[z][pmid[z] < L2.tropopause] = np.nan
            mv = np.nansum(pc, axis=0)
            mv[np.isnan(L2.vcd)] = np.nan
            L2.ctm_vcd, L2.ctm_time_at_sat = mv, tc[ci]
            L2.old_amf, L2.new_amf = np.empty((1)), np.empty((1))
            continue
        new_amf = np.full_like(L2.vcd, np.nan)
        mv = np.full_like(L2.vcd, np.nan)
        for i in range(L2.vcd.shape[0]):
            for j in range(L2.vcd.shape[1]):
                if np.isnan(L2.vcd[i, j]):
                    continue
                new_amf[i, j], mv[i, j] = amf_pixel(L2.pressure_mid[:, i, j], L2.scattering_weights[:, i, j], pmid[:, i, j],
                                                    pc[:, i, j], L2.tropopause[i, j] if has_trop else None)
        L2.old_amf = L2.amf
        new_amf[np.isnan(L2.vcd)] = np.nan
        L2.new_amf = new_amf
        with np.errstate(all="ignore"):
            L2.vcd = (L2.amf * L2.vcd) / new_amf
        mv[np.isnan(L2.vcd)] = np.nan
        mv[np.isinf(L2.vcd)] = np.nan
        L2.ctm_vcd, L2.ctm_time_at_sat = mv, tc[ci]
    return sat_data


# --------------------------------------------------------------------------------------------
# ak_conv_mopitt.py / ak_conv_gosat.py  (the satellite_opt counterpart of amf_recal, driver.py:46-51)
# --------------------------------------------------------------------------------------------
def air_partial_column(deltap):
    """ak_conv_mopitt.py:66."""
    return deltap / 9.80665 / 28.97e-3 * 6.02214076e23 * 1e-4 * 1e-15 * 100.0


def mopitt_pixel(ctm_p, ctm_prof, ctm_air, sat_p, ak, ap_prof, ap_col, ap_surf):
    """One pixel of ak_conv_mopitt.py:120-138 -> (model_VCD, model_xcol)."""
    f = _interp1d(np.log(ctm_p), ctm_prof, fill_value=np.nan, bounds_error=False)
    with np.errstate(all="ignore"):
        xi = f(np.log(sat_p))
        prof_part = ap_col + np.nansum(ak[1:] * (np.log10(xi) - np.log10(ap_prof)))
        surf_part = ak[0] * (np.log10(ctm_prof[0]) - np.log10(ap_surf))
        v = prof_part + surf_part
        return v, 1e6 * v / np.nansum(ctm_air)


def gosat_pixel(ctm_p, ctm_prof, sat_p, ak, ap_prof, pw):
    """One pixel of ak_conv_gosat.py:124-135 -> model_xcol."""
    f = _interp1d(np.log(ctm_p), ctm_prof, fill_value="extrapolate")
    with np.errstate(all="ignore"):
        xi = f(np.log(sat_p))
        t = (ap_prof + (xi - ap_prof) * ak) * pw
        t[t <= 0] = np.nan
        return np.nansum(t)


def ak_conv(ctm_data, sat_data, sensor):
    """``ak_conv_mopitt`` (ak_conv_mopitt.py:8-149) / ``ak_conv_gosat`` (ak_conv_gosat.py:8-146).  Mutates and
    returns ``sat_data`` like the reference."""
    tc = np.array([_flat_time(t) for rec in ctm_data for t in rec.time])
    for L2 in sat_data:
        if L2 is None:
            continue
        ts = L2.time.year * 10000 + L2.time.month * 100 + L2.time.day
        ci = int(np.argmin(np.abs(ts - tc))) if not ctm_data[0].averaged else 0
        rec = ctm_data[ci]                                  # (the reference indexes records with the time-slot index)
        if rec.ctmtype in ("ECCOH", "FREE"):
            pmid, prof, delp = rec.pressure_mid.squeeze(), rec.gas_profile.squeeze(), rec.delta_p.squeeze()
        elif rec.ctmtype == "GMI":
            with np.errstate(all="ignore"):
                pmid = np.nanmean(rec.pressure_mid, axis=0).squeeze()
                prof = np.nanmean(rec.gas_profile, axis=0).squeeze()
                delp = np.nanmean(rec.delta_p, axis=0).squeeze()
        else:
            raise NameError("ctm_mid_pressure is not defined for ctmtype " + str(rec.ctmtype))
        air = air_partial_column(delp)
        if L2.ctm_upscaled_needed:
            coord = {"Longitude": L2.longitude_center, "Latitude": L2.latitude_center}
            thr = np.sqrt(np.abs(coord["Longitude"][0, 0] - coord["Longitude"][0, 1]) ** 2 +
                          np.abs(coord["Latitude"][0, 0] - coord["Latitude"][1, 0]) ** 2)
            clon, clat = ctm_data[0].longitude, ctm_data[0].latitude
            gs = np.sqrt(np.abs(clon[0, 0] - clon[0, 1]) ** 2 + np.abs(clat[0, 0] - clat[1, 0]) ** 2)
            up = lambda c: np.stack([upscaler(clon, clat, c[z], coord, gs, thr)[2] for z in range(c.shape[0])])   # noqa: E731
            pmid, prof, air = up(pmid), up(prof), up(air)
        key = L2.vcd if sensor == "MOPITT" else L2.x_col
        mv = np.full_like(L2.vcd, np.nan)
        mx = np.full_like(L2.vcd, np.nan)
        for i in range(key.shape[0]):
            for j in range(key.shape[1]):
                if np.isnan(key[i, j]):
                    continue
                if sensor == "MOPITT":
                    mv[i, j], mx[i, j] = mopitt_pixel(pmid[:, i, j], prof[:, i, j], air[:, i, j], L2.pressure_mid[:, i, j],
                                                      L2.averaging_kernels[:, i, j], L2.apriori_profile[:, i, j],
                                                      L2.aprior_column[i, j], L2.apriori_surface[i, j])
                else:
                    mx[i, j] = gosat_pixel(pmid[:, i, j], prof[:, i, j], L2.pressure_mid[:, i, j],
                                           L2.averaging_kernels[:, i, j], L2.apriori_profile[:, i, j], L2.pressure_weight[:, i, j])
        if sensor == "MOPITT":
            mv[np.isnan(L2.vcd)] = np.nan
            mv[np.isinf(L2.vcd)] = np.nan
        else:
            mx[np.isinf(L2.x_col)] = np.nan
            mx[np.isnan(L2.x_col)] = np.nan
        L2.ctm_vcd, L2.ctm_xcol, L2.ctm_time_at_sat = mv, mx, tc[ci]
    return sat_data


def pwv_calculator(ctm_data, sat_data):
    """``pwv_calculator`` (pwv_cal.py:7-101): model precipitable water for SSMIS.  Mutates and returns ``sat_data``."""
    tc = np.array([_flat_time(t) for rec in ctm_data for t in rec.time])
    for L2 in sat_data:
        if L2 is None:
            continue
        ts = L2.time.year * 10000 + L2.time.month * 100 + L2.time.day
        ci = int(np.argmin(np.abs(ts - tc))) if not ctm_data[0].averaged else 0
        rec = ctm_data[ci]
        if rec.ctmtype in ("ECCOH", "FREE"):
            delp, prof = rec.delta_p.squeeze(), rec.gas_profile.squeeze()
        elif rec.ctmtype == "GMI":
            with np.errstate(all="ignore"):
                prof = np.nanmean(rec.gas_profile, axis=0).squeeze()
                delp = np.nanmean(rec.delta_p, axis=0).squeeze()
        else:
            raise NameError("ctm_deltap is not defined for ctmtype " + str(rec.ctmtype))
        pc = delp * prof / 9.80665 / 10000.0
        if L2.ctm_upscaled_needed:
            coord = {"Longitude": L2.longitude_center, "Latitude": L2.latitude_center}
            thr = np.sqrt(np.abs(coord["Longitude"][0, 0] - coord["Longitude"][0, 1]) ** 2 +
                          np.abs(coord["Latitude"][0, 0] - coord["Latitude"][1, 0]) ** 2)
            clon, clat = ctm_data[0].longitude, ctm_data[0].latitude
            gs = np.sqrt(np.abs(clon[0, 0] - clon[0, 1]) ** 2 + np.abs(clat[0, 0] - clat[1, 0]) ** 2)
            pc = np.stack([upscaler(clon, clat, pc[z], coord, gs, thr)[2] for z in range(pc.shape[0])])
        pwv = np.nansum(pc / 1000.0, axis=0).squeeze()
        pwv[np.isnan(L2.vcd)] = np.nan
        pwv[np.isinf(L2.vcd)] = np.nan
        L2.ctm_vcd = pwv
    return sat_data


# --------------------------------------------------------------------------------------------
# Dense Gaussian-B OI (north-star extension; NO reference counterpart -> parity unpinned)
# --------------------------------------------------------------------------------------------
def unit_vectors(lat_deg, lon_deg):
    la = np.deg2rad(np.asarray(lat_deg, dtype=np.float64))
    lo = np.deg2rad(np.asarray(lon_deg, dtype=np.float64))
    return np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)], axis=-1)


def gaussian_corr(pa, pb, L_km):
    """C = exp(-c^2 R^2 / (2 L^2)), c = chord length between unit vectors (positive definite on
    the sphere, unlike a Gaussian of great-circle distance)."""
    d2 = ((pa[:, None, :] - pb[None, :, :]) ** 2).sum(axis=-1)
    return np.exp(-d2 * (EARTH_RADIUS_KM / L_km) ** 2 / 2.0)


def dense_oi(grid_lat, grid_lon, Xa, Sa, obs_lat, obs_lon, obs_cell, obs_y, obs_var, L_km,
             scale=1.0, want_error=False, chunk=4096):
    """x_a = x_b + B H^T (H B H^T + R)^-1 (y - H x_b), B = s D^1/2 C D^1/2, D = diag(Sa),
    C Gaussian in chord distance, H = selection of ``obs_cell`` (flat index), R = diag(obs_var).
    Returns dict(xa, inc, z, ak_obs, err) in float64."""
    xa = np.asarray(Xa, dtype=np.float64).ravel()
    sb = np.sqrt(scale * np.asarray(Sa, dtype=np.float64).ravel())
    pg = unit_vectors(np.ravel(grid_lat), np.ravel(grid_lon))
    po = unit_vectors(obs_lat, obs_lon)
    so = sb[obs_cell]
    S = gaussian_corr(po, po, L_km) * so[:, None] * so[None, :]
    S[np.diag_indices_from(S)] += obs_var
    d = obs_y - xa[obs_cell]
    cf = _sla.cho_factor(S, lower=True)
    z = _sla.cho_solve(cf, d)
    inc = np.empty_like(xa)
    err = np.full_like(xa, np.nan)
    w = so * z
    for i0 in range(0, xa.size, chunk):
        i1 = min(i0 + chunk, xa.size)
        Cg = gaussian_corr(pg[i0:i1], po, L_km)
        inc[i0:i1] = sb[i0:i1] * (Cg @ w)
        if want_error:
            BHt = Cg * sb[i0:i1, None] * so[None, :]                  # rows of B H^T
            V = _sla.solve_triangular(cf[0], BHt.T, lower=True)       # L^-1 (H B)
            err[i0:i1] = np.sqrt(np.maximum(sb[i0:i1] ** 2 - (V * V).sum(axis=0), 0.0))
    Sinv_diag = None
    ak_obs = None
    if want_error:
        Linv = _sla.solve_triangular(cf[0], np.eye(S.shape[0]), lower=True)
        Sinv_diag = (Linv * Linv).sum(axis=0)
        ak_obs = 1.0 - obs_var * Sinv_diag                            # diag(K H) at the obs cells
    return {"xa": xa + inc, "inc": inc, "z": z, "ak_obs": ak_obs, "err": err, "d": d}
