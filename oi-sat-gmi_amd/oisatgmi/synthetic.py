"""Seeded synthetic inputs for the optimal-interpolation hot path.

The reference ships no test data, so every parity vector and every
bench workload is built from these generators:

* global cell-centred grids  lat = -90+d/2 .. 90-d/2,  lon = -180+d/2 .. 180-d/2
* ``Xa``: smooth positive NO2-like column field in ~[0.2, 10] (x1e15 molec/cm2, the range the
  reference plots, report.py:132-139)
* truth = Xa * (1 + 0.3 g), g smooth ~N(0,1);  y = truth + N(0, sigma_o^2), sigma_o ~ U(0.1, 1)
  about 1 % of the observations are pushed negative to exercise the ``Y[Y<0]=0`` clamp
  (optimal_interpolation.py:14)
* ``Sa = (Xa * ctm_error/100)^2`` and ``So = sigma_o^2`` exactly as driver.py:110-111 builds them

Everything is ``numpy.random.default_rng(seed)`` with ``seed = 1000*config + case``.
This module is data generation only: it never touches the GPU and never imports the oracle.
"""
from __future__ import annotations

import datetime as _dt
from dataclasses import dataclass

import numpy as np

EARTH_RADIUS_KM = 6371.0

# BASELINE config 5: the three control_*.yml shapes.  ctm_error from
# run/control_omino2.yml:23, control_omihcho.yml, control_omio3.yml; value ranges = the plotting ranges of
# report.py:124-146 (x1e15 molec/cm2, O3 in DU); OMI O3 observation error = 4 % of the column (reader.py:1035).
SPECIES = {
    "NO2": dict(ctm_error=50.0, value_range=(0.2, 10.0), base=1.0, amp=4.0, obs_err=(0.1, 1.0), rel_obs_err=None),
    "HCHO": dict(ctm_error=50.0, value_range=(0.5, 20.0), base=3.0, amp=8.0, obs_err=(0.5, 4.0), rel_obs_err=None),
    "O3": dict(ctm_error=10.0, value_range=(200.0, 500.0), base=250.0, amp=150.0, obs_err=None, rel_obs_err=0.04),
}


def global_grid(ny: int, nx: int):
    """Cell-centred global lat/lon mesh, shape (ny, nx), float64 degrees."""
    dlat = 180.0 / ny
    dlon = 360.0 / nx
    lat = -90.0 + dlat / 2 + dlat * np.arange(ny)
    lon = -180.0 + dlon / 2 + dlon * np.arange(nx)
    lon2, lat2 = np.meshgrid(lon, lat)
    return lat2, lon2


def _smooth_unit_field(rng, lat2, lon2, nmodes=12, kmax=5):
    """Sum of random low-order lon/lat waves, normalised to zero mean / unit variance."""
    lam = np.deg2rad(lon2)
    phi = np.deg2rad(lat2)
    g = np.zeros_like(lat2)
    for _ in range(nmodes):
        k1 = rng.integers(1, kmax + 1)
        k2 = rng.integers(1, kmax + 1)
        p1 = rng.uniform(0, 2 * np.pi)
        p2 = rng.uniform(0, 2 * np.pi)
        g += rng.normal() * np.sin(k1 * lam + p1) * np.cos(k2 * phi + p2)
    g -= g.mean()
    g /= g.std()
    return g


def background_field(rng, lat2, lon2, nspots=24, lo=0.2, hi=10.0, base=1.0, amp=4.0):
    """Smooth positive field: base + amp*(1+sin3l*cos2p)/2 + lognormal hot spots, clipped."""
    lam = np.deg2rad(lon2)
    phi = np.deg2rad(lat2)
    xa = base + amp * (1.0 + np.sin(3 * lam) * np.cos(2 * phi)) / 2.0
    for _ in range(nspots):
        clat = rng.uniform(-60, 60)
        clon = rng.uniform(-180, 180)
        w = rng.uniform(2.0, 8.0)
        a = rng.lognormal(mean=0.5, sigma=0.6)
        dlon = (lon2 - clon + 180.0) % 360.0 - 180.0
        xa += a * np.exp(-((lat2 - clat) ** 2 + dlon ** 2) / (2 * w * w))
    return np.clip(xa, lo, hi)


@dataclass
class DiagCase:
    """Gridded (Tier-A) OI inputs in the reference's own layout: four (ny, nx) float64 fields."""
    lat: np.ndarray
    lon: np.ndarray
    Xa: np.ndarray
    Y: np.ndarray          # NaN where unobserved
    Sa: np.ndarray
    So: np.ndarray         # NaN where unobserved
    truth: np.ndarray
    sat_err: np.ndarray    # sigma_o, NaN where unobserved


def diag_case(ny: int, nx: int, nobs: int, seed: int, ctm_error: float = 50.0,
              neg_frac: float = 0.01, value_range=(0.2, 10.0), base=1.0, amp=4.0,
              rel_obs_err: float | None = None) -> DiagCase:
    """Config 1/2/3-style gridded case: ``nobs`` distinct random cells observed."""
    rng = np.random.default_rng(seed)
    lat2, lon2 = global_grid(ny, nx)
    xa = background_field(rng, lat2, lon2, lo=value_range[0], hi=value_range[1], base=base, amp=amp)
    g = _smooth_unit_field(rng, lat2, lon2)
    truth = xa * (1.0 + 0.3 * g)
    n = ny * nx
    nobs = min(nobs, n)
    cells = rng.choice(n, size=nobs, replace=False)
    if rel_obs_err is None:
        sig = rng.uniform(0.1, 1.0, size=nobs)
    else:                                   # O3-style: sigma_o = 4 % of y (reader.py:1035)
        sig = rel_obs_err * np.abs(truth.ravel()[cells])
    y = truth.ravel()[cells] + rng.normal(size=nobs) * sig
    nneg = int(round(neg_frac * nobs))
    if nneg:
        y[rng.choice(nobs, size=nneg, replace=False)] *= -0.1
    Y = np.full(n, np.nan)
    So = np.full(n, np.nan)
    E = np.full(n, np.nan)
    Y[cells] = y
    So[cells] = sig ** 2
    E[cells] = sig
    Sa = (xa * ctm_error / 100.0) ** 2
    return DiagCase(lat2, lon2, xa, Y.reshape(ny, nx), Sa, So.reshape(ny, nx), truth, E.reshape(ny, nx))


@dataclass
class PointObsCase:
    """Dense (Tier-B) OI inputs: gridded background + scattered point observations."""
    lat: np.ndarray        # (ny, nx)
    lon: np.ndarray
    Xa: np.ndarray
    Sa: np.ndarray
    obs_lat: np.ndarray    # (m,)
    obs_lon: np.ndarray
    obs_y: np.ndarray
    obs_var: np.ndarray    # sigma_o^2
    truth: np.ndarray


def point_obs_case(ny: int, nx: int, nobs: int, seed: int, ctm_error: float = 50.0,
                   swaths: bool = False, cloud_frac: float = 0.6, species: str = "NO2") -> PointObsCase:
    """Random (lat, lon) observations.  ``swaths=True`` gives the OMI-NO2-style layout of
    BASELINE config 3: 14-15 stripes ~2600 km wide with ``cloud_frac`` of the pixels dropped.
    ``species`` picks one of the ``SPECIES`` parameter sets (config 5); the default is the NO2 case."""
    sp = SPECIES[species]
    if species != "NO2":
        ctm_error = sp["ctm_error"]
    rng = np.random.default_rng(seed)
    lat2, lon2 = global_grid(ny, nx)
    xa = background_field(rng, lat2, lon2, lo=sp["value_range"][0], hi=sp["value_range"][1], base=sp["base"], amp=sp["amp"])
    g = _smooth_unit_field(rng, lat2, lon2)
    truth = xa * (1.0 + 0.3 * g)
    if not swaths:
        # area-uniform on the sphere, kept off the poles where the lat/lon grid degenerates
        u = rng.uniform(np.sin(np.deg2rad(-85)), np.sin(np.deg2rad(85)), size=nobs)
        olat = np.rad2deg(np.arcsin(u))
        olon = rng.uniform(-180, 180, size=nobs)
    else:
        nst = 14 + int(seed % 2)
        want = int(nobs / max(1e-6, 1.0 - cloud_frac)) + nst
        per = want // nst
        la, lo = [], []
        half_w = 2600.0 / 2 / 111.2            # degrees of longitude at the equator
        for s in range(nst):
            c = -180.0 + (s + 0.5) * 360.0 / nst
            t = rng.uniform(-80, 80, size=per)
            off = rng.uniform(-half_w, half_w, size=per) / np.maximum(np.cos(np.deg2rad(t)), 0.2)
            la.append(t)
            lo.append(((c + off + 0.3 * t) + 180.0) % 360.0 - 180.0)   # inclined ground track
        olat = np.concatenate(la)
        olon = np.concatenate(lo)
        keep = rng.uniform(size=olat.size) > cloud_frac
        olat, olon = olat[keep][:nobs], olon[keep][:nobs]
        nobs = olat.size
    # truth at the observation = truth of the containing cell
    iy = np.clip(np.floor((olat + 90.0) / (180.0 / ny)).astype(np.int64), 0, ny - 1)
    ix = np.clip(np.floor((olon + 180.0) / (360.0 / nx)).astype(np.int64), 0, nx - 1)
    if sp["rel_obs_err"] is None:
        sig = rng.uniform(sp["obs_err"][0], sp["obs_err"][1], size=nobs)
    else:
        sig = sp["rel_obs_err"] * np.abs(truth[iy, ix])
    y = truth[iy, ix] + rng.normal(size=nobs) * sig
    Sa = (xa * ctm_error / 100.0) ** 2
    return PointObsCase(lat2, lon2, xa, Sa, olat, olon, y, sig ** 2, truth)


def granule_stack(ny: int, nx: int, k: int, seed: int, year: int = 2019, month: int = 6,
                  with_none: bool = True, coverage: float = 0.35):
    """``k`` regridded daily granules for the averaging stage, as ``satellite_amf`` records on the
    model grid (the objects ``averaging()`` consumes, averaging.py:72-90).  Each granule covers a
    random ``coverage`` of the cells (NaN elsewhere); a few +/-inf values are planted in vcd and
    uncertainty (averaging.py:92, :19 special-case them) and, if ``with_none``, ``None`` entries
    are interleaved the way failed granules are (averaging.py:73-74)."""
    from .config import satellite_amf  # local import: config is tiny and has no GPU dependency
    rng = np.random.default_rng(seed)
    lat2, lon2 = global_grid(ny, nx)
    base = background_field(rng, lat2, lon2)
    out = []
    for d in range(k):
        mask = rng.uniform(size=(ny, nx)) < coverage
        vcd = np.where(mask, base * (1 + 0.2 * rng.normal(size=(ny, nx))), np.nan)
        unc = np.where(mask, rng.uniform(0.1, 1.0, size=(ny, nx)), np.nan)
        ctm = np.where(mask, base * (1 + 0.05 * rng.normal(size=(ny, nx))), np.nan)
        new_amf = np.where(mask, rng.uniform(0.5, 2.5, size=(ny, nx)), np.nan)
        old_amf = np.where(mask, rng.uniform(0.5, 2.5, size=(ny, nx)), np.nan)
        # plant specials in observed cells
        idx = np.argwhere(mask)
        if idx.shape[0] > 8:
            for t, (i, j) in enumerate(idx[rng.choice(idx.shape[0], size=4, replace=False)]):
                if t % 2 == 0:
                    vcd[i, j] = np.inf if t == 0 else -np.inf
                else:
                    unc[i, j] = np.inf
        when = _dt.datetime(year, month, 1 + (d % 28), 13, 30) + _dt.timedelta(minutes=7 * d)
        out.append(satellite_amf(vcd, np.empty((1)), when, np.empty((1)), lat2, lon2, [], [],
                                 unc, [], np.empty((1)), np.empty((1)), False, ctm, when,
                                 old_amf, new_amf))
        if with_none and d % 3 == 1:
            out.append(None)
    # one granule from the previous month that the (year, month) filter must drop (averaging.py:77)
    prev = _dt.datetime(year, month, 1, 12, 0) - _dt.timedelta(days=3)
    g0 = out[0]
    out.append(satellite_amf(g0.vcd * 100.0, g0.amf, prev, g0.tropopause, lat2, lon2, [], [],
                             g0.uncertainty * 100.0, [], g0.pressure_mid, g0.scattering_weights,
                             False, g0.ctm_vcd * 100.0, prev, g0.old_amf, g0.new_amf))
    return out


def swath_granule(seed: int, nscan: int = 240, npix: int = 60, lat0: float = -20.0, lat1: float = 40.0,
                  lon_c: float = 10.0, width_deg: float = 24.0, hole: bool = True):
    """One jittered, inclined L2 swath (``satellite_amf``) for the regridding stage
    (what ``interpolator()`` receives from the readers, e.g. reader.py:894-903).  A block of
    pixels gets quality_flag below threshold (values masked, points kept: interpolator.py:126-132)."""
    from .config import satellite_amf
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 1.0, nscan)[:, None]
    s = np.linspace(-0.5, 0.5, npix)[None, :]
    lat = lat0 + (lat1 - lat0) * t + 0.8 * s + rng.normal(scale=0.01, size=(nscan, npix))
    lon = lon_c + 8.0 * (t - 0.5) + width_deg * s + rng.normal(scale=0.01, size=(nscan, npix))
    vcd = 2.0 + np.sin(np.deg2rad(4 * lon)) * np.cos(np.deg2rad(3 * lat)) + 0.05 * rng.normal(size=lat.shape)
    amf = 1.0 + 0.5 * np.cos(np.deg2rad(lat)) + 0.01 * rng.normal(size=lat.shape)
    unc = rng.uniform(0.1, 0.6, size=lat.shape)
    qf = np.ones_like(lat)
    if hole:
        qf[nscan // 3: nscan // 3 + nscan // 8, npix // 4: npix // 2] = 0.3
    when = _dt.datetime(2019, 6, 15, 13, 45)
    return satellite_amf(vcd, amf, when, np.empty((1)), lat, lon, [], [], unc, qf,
                         np.empty((1)), np.empty((1)), False, [], [], [], [])


def swath_level_granule(seed: int, kind: str = "amf", nz: int = 3, nscan: int = 120, npix: int = 40, lat0: float = -5.0,
                        lat1: float = 25.0, lon_c: float = 10.0, width_deg: float = 10.0):
    """One small L2 swath that carries the per-level cubes ``interpolator()`` regrids level by level
    (interpolator.py:191-283): ``kind='amf'`` -> ``satellite_amf`` with scattering weights + pressure_mid (OMI-NO2
    style, reader.py:874-883) and a tropopause field; ``'MOPITT'`` / ``'GOSAT'`` -> ``satellite_opt`` with averaging
    kernels (nz+1 rows for MOPITT, nz for GOSAT), a-priori profile, pressure weights (GOSAT), a-priori column /
    surface, surface pressure and x_col.  Cubes are level-major (nz, nscan, npix); a block of pixels is flagged bad."""
    from .config import satellite_amf, satellite_opt
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 1.0, nscan)[:, None]
    s = np.linspace(-0.5, 0.5, npix)[None, :]
    lat = lat0 + (lat1 - lat0) * t + 0.6 * s + rng.normal(scale=0.01, size=(nscan, npix))
    lon = lon_c + 4.0 * (t - 0.5) + width_deg * s + rng.normal(scale=0.01, size=(nscan, npix))
    shape = lat.shape
    vcd = 2.0 + np.sin(np.deg2rad(6 * lon)) * np.cos(np.deg2rad(5 * lat)) + 0.05 * rng.normal(size=shape)
    unc = rng.uniform(0.1, 0.6, size=shape)
    qf = np.ones(shape)
    qf[nscan // 3: nscan // 3 + nscan // 8, npix // 4: npix // 2] = 0.3
    when = _dt.datetime(2019, 6, 15, 13, 45)
    smooth = 1.0 + 0.3 * np.sin(np.deg2rad(7 * lon + 3 * lat))
    pmid = np.linspace(950.0, 120.0, nz)[:, None, None] * (1.0 + 0.01 * rng.normal(size=(nz,) + shape))
    if kind == "amf":
        amf = 1.0 + 0.5 * np.cos(np.deg2rad(lat)) + 0.01 * rng.normal(size=shape)
        sw = (np.linspace(0.4, 1.8, nz)[:, None, None] * smooth[None]).astype(np.float32)      # float32 cubes, as the readers hand over
        trop = 200.0 + 50.0 * np.cos(np.deg2rad(3 * lat)) + rng.normal(size=shape)
        return satellite_amf(vcd, amf, when, trop, lat, lon, [], [], unc, qf, pmid.astype(np.float32), sw, False, [], [], [], [])
    nak = nz + 1 if kind == "MOPITT" else nz
    ak = np.linspace(0.2, 1.2, nak)[:, None, None] * smooth[None] + 0.01 * rng.normal(size=(nak,) + shape)
    apro = np.linspace(90.0, 40.0, nz)[:, None, None] * smooth[None]
    pw = np.empty((1))
    if kind == "GOSAT":
        pw = np.full((nz,) + shape, 1.0 / nz) * (1.0 + 0.05 * rng.normal(size=(nz,) + shape))
    return satellite_opt(vcd, when, [], np.empty((1)), lat, lon, [], [], unc, qf, pmid, ak, False, [], [], [],
                         18.0 + 0.2 * smooth, apro, 1000.0 - 20.0 * smooth, 80.0 * smooth,
                         1800.0 + 40.0 * smooth + rng.normal(size=shape), pw, kind)


def lattice_l3_granule(seed: int, sensor: str = "MOPITT", nz: int = 3, lat0: float = -19.5, lat1: float = 19.5,
                       lon0: float = -29.5, lon1: float = 29.5, step: float = 1.0):
    """A gridded (level-3) ``satellite_opt`` record as reader.py:1150-1203 hands MOPITT MOP03 files to
    ``interpolator``: cell centres on an exact ``step``-degree lattice read as float32 and laid out lon-major
    (``meshgrid`` + transpose, reader.py:1157-1160), quality flag all ones (``flag_thresh=0.0`` at the call site,
    reader.py:1209-1211), NaN holes in the retrieved column.  With ``grid_size = step`` every fine-grid node the
    reference builds (interpolator.py:141-143) sits exactly between lattice centres, and every model centre at x.5
    exactly between fine nodes: the exact nearest-neighbour ties of SURVEY section 8(a) row a7."""
    from .config import satellite_opt
    rng = np.random.default_rng(seed)
    lon1d = np.arange(lon0, lon1 + 1e-9, step).astype(np.float32)
    lat1d = np.arange(lat0, lat1 + 1e-9, step).astype(np.float32)
    lon, lat = np.meshgrid(lon1d, lat1d)
    lon, lat = np.transpose(lon), np.transpose(lat)
    shape = lat.shape
    vcd = 2.0 + np.sin(np.deg2rad(6 * lon)) * np.cos(np.deg2rad(5 * lat)) + 0.05 * rng.normal(size=shape)
    vcd[rng.uniform(size=shape) < 0.03] = np.nan
    unc = rng.uniform(0.1, 0.6, size=shape).astype(np.float32)
    smooth = 1.0 + 0.3 * np.sin(np.deg2rad(7 * lon + 3 * lat))
    pmid = np.linspace(950.0, 120.0, nz)[:, None, None] * np.ones((nz,) + shape)
    nak = nz + 1 if sensor == "MOPITT" else nz
    ak = np.linspace(0.2, 1.2, nak)[:, None, None] * smooth[None] + 0.01 * rng.normal(size=(nak,) + shape)
    apro = np.linspace(90.0, 40.0, nz)[:, None, None] * smooth[None]
    pw = np.empty((1))
    if sensor == "GOSAT":
        pw = np.full((nz,) + shape, 1.0 / nz) * (1.0 + 0.05 * rng.normal(size=(nz,) + shape))
    when = _dt.datetime(2019, 6, 15, 10, 30)
    return satellite_opt(vcd, when, [], np.empty((1)), lat, lon, [], [], unc, np.ones_like(vcd), pmid, ak, [], [], [], [],
                         18.0 + 0.2 * smooth, apro, 1000.0 - 20.0 * smooth, 80.0 * smooth,
                         1800.0 + 40.0 * smooth + rng.normal(size=shape), pw, sensor)


def regional_ctm_grid(lat0: float, lat1: float, lon0: float, lon1: float, dlat: float, dlon: float):
    """Model-grid coordinate dict in the layout the readers hand to ``interpolator``
    (``{'Latitude': 2-D, 'Longitude': 2-D}``, interpolator.py:117-118)."""
    lat = np.arange(lat0, lat1 + 1e-9, dlat)
    lon = np.arange(lon0, lon1 + 1e-9, dlon)
    lon2, lat2 = np.meshgrid(lon, lat)
    return {"Latitude": lat2, "Longitude": lon2}


def ctm_days(ny: int, nx: int, nz: int, ndays: int, seed: int, averaged: bool = False, dtype=np.float32,
             lat0=-30.0, lat1=30.0, lon0=-40.0, lon1=40.0, year=2019, month=6):
    """Model records as ``amf_recal`` consumes them (amf_recal.py:39-49,:124-131): one ``ctm_model`` per day
    with 8 three-hourly slots (gas_profile, pressure_mid, delta_p of shape (8, nz, ny, nx)), or, with
    ``averaged=True``, a single record holding the month's mean diurnal cycle."""
    from .config import ctm_model
    rng = np.random.default_rng(seed)
    lat = np.linspace(lat0, lat1, ny)
    lon = np.linspace(lon0, lon1, nx)
    lon2, lat2 = np.meshgrid(lon, lat)
    # pressure levels: surface (index 0) to top, hPa, with a little horizontal variation
    edges = np.linspace(1000.0, 50.0, nz + 1)
    pm = 0.5 * (edges[:-1] + edges[1:])
    dp = (edges[:-1] - edges[1:])
    out = []
    for d in range(1 if averaged else ndays):
        times = [_dt.datetime(year, month, 1 + d, 3 * h, 0) for h in range(8)]
        wob = 1.0 + 0.01 * rng.normal(size=(8, 1, ny, nx))
        pmid = (pm[None, :, None, None] * wob).astype(dtype)
        delp = (dp[None, :, None, None] * wob).astype(dtype)
        prof = (rng.lognormal(mean=-1.0, sigma=0.5, size=(8, nz, ny, nx)) * np.exp(-np.arange(nz) / 6.0)[None, :, None, None]).astype(dtype)
        out.append(ctm_model(lat2, lon2, times, prof, pmid, np.zeros_like(pmid), delp, "GMI", averaged))
    return out


def amf_granules(ctm, nzs: int, k: int, seed: int, with_sw: bool = True, with_trop: bool = True):
    """``k`` regridded granules on the model grid (``ctm_upscaled_needed=False``) carrying what the AMF
    recalculation needs: scattering weights and their pressure grid (level-major), AMF, tropopause."""
    from .config import satellite_amf
    rng = np.random.default_rng(seed)
    lat2, lon2 = ctm[0].latitude, ctm[0].longitude
    ny, nx = lat2.shape
    out = []
    for g in range(k):
        when = _dt.datetime(2019, 6, 1 + g % 2, 13, 20 + g)
        vcd = rng.uniform(0.5, 8.0, size=(ny, nx))
        vcd[rng.uniform(size=vcd.shape) < 0.2] = np.nan
        amf = rng.uniform(0.6, 2.5, size=(ny, nx))
        unc = rng.uniform(0.1, 1.0, size=(ny, nx))
        if with_sw:
            # satellite levels top-down or bottom-up, alternating: interp1d must sort them
            p = np.linspace(1020.0, 20.0, nzs)[:, None, None] * (1 + 0.005 * rng.normal(size=(nzs, ny, nx)))
            if g % 2:
                p = p[::-1].copy()
            sw = rng.uniform(0.2, 2.0, size=(nzs, ny, nx))
        else:
            p, sw = np.empty((1)), np.empty((1))
        trop = rng.uniform(80.0, 300.0, size=(ny, nx)) if with_trop else np.empty((1))
        out.append(satellite_amf(vcd, amf, when, trop, lat2, lon2, [], [], unc, [], p, sw, False, [], [], [], []))
    out.insert(1, None)
    return out


def ctm_monthly(ny: int, nx: int, nz: int, nmonths: int, seed: int, ctmtype: str = "ECCOH", dtype=np.float32,
                lat0=-30.0, lat1=30.0, lon0=-40.0, lon1=40.0, year=2019, month0=5):
    """Model records as the averaging-kernel convolution consumes them (ak_conv_mopitt.py:60-77): ``ECCOH`` / ``FREE``
    records hold one monthly mean, cubes (1, nz, ny, nx); ``GMI`` records hold 8 time slots that are nan-averaged."""
    from .config import ctm_model
    rng = np.random.default_rng(seed)
    lat = np.linspace(lat0, lat1, ny)
    lon = np.linspace(lon0, lon1, nx)
    lon2, lat2 = np.meshgrid(lon, lat)
    edges = np.linspace(1000.0, 60.0, nz + 1)
    pm = 0.5 * (edges[:-1] + edges[1:])
    dp = (edges[:-1] - edges[1:])
    nt = 8 if ctmtype == "GMI" else 1
    out = []
    for mth in range(nmonths):
        times = [_dt.datetime(year, month0 + mth, 1, 3 * h if nt > 1 else 0, 0) for h in range(nt)]
        wob = 1.0 + 0.01 * rng.normal(size=(nt, 1, ny, nx))
        pmid = (pm[None, :, None, None] * wob).astype(dtype)
        delp = (dp[None, :, None, None] * wob).astype(dtype)
        prof = (rng.lognormal(mean=4.0, sigma=0.3, size=(nt, nz, ny, nx)) * np.exp(-np.arange(nz) / 9.0)[None, :, None, None]).astype(dtype)
        if nt > 1:
            prof[3, :, 1, 2] = np.nan                        # a missing time slot: nanmean skips it
        out.append(ctm_model(lat2, lon2, times, prof, pmid, np.zeros_like(pmid), delp, ctmtype, False))
    return out


def opt_granules(ctm, nzs: int, k: int, seed: int, sensor: str = "MOPITT"):
    """``k`` gridded optimal-estimation granules (``satellite_opt``) on the model grid for ``conv_ak``:
    MOPITT carries nzs profile levels + 1 surface averaging-kernel row and log10 a-priori terms
    (ak_conv_mopitt.py:118-138); GOSAT carries nzs levels, pressure weights and XCH4 (ak_conv_gosat.py:118-135)."""
    from .config import satellite_opt
    rng = np.random.default_rng(seed)
    lat2, lon2 = ctm[0].latitude, ctm[0].longitude
    ny, nx = lat2.shape
    out = []
    for g in range(k):
        when = _dt.datetime(2019, 5 + g % 2, 3 + g, 10, 30)
        # satellite levels overshoot the model's pressure range at both ends: NaN (MOPITT) / extrapolation (GOSAT) there
        p = np.linspace(1030.0, 40.0, nzs)[:, None, None] * (1 + 0.004 * rng.normal(size=(nzs, ny, nx)))
        if g % 2:
            p = p[::-1].copy()
        ap = rng.lognormal(mean=4.0, sigma=0.2, size=(nzs, ny, nx))
        vcd = rng.uniform(1.0, 3.0, size=(ny, nx))
        xcol = rng.uniform(1700.0, 1900.0, size=(ny, nx))
        for a in (vcd, xcol):
            a[rng.uniform(size=a.shape) < 0.15] = np.nan
            a[2, 3] = np.inf
        unc = rng.uniform(0.05, 0.3, size=(ny, nx))
        if sensor == "MOPITT":
            ak = rng.normal(scale=0.3, size=(nzs + 1, ny, nx))
            pw = np.empty((1))
        else:
            ak = rng.uniform(0.2, 1.4, size=(nzs, ny, nx))
            pw = rng.dirichlet(np.ones(nzs), size=(ny, nx)).transpose(2, 0, 1).copy()
            pw[0, 1, 1] = -pw[0, 1, 1]                      # a non-positive term is dropped (ak_conv_gosat.py:133)
        out.append(satellite_opt(vcd, when, [], np.empty((1)), lat2, lon2, [], [], unc, [], p, ak, False, [], [], [],
                                 rng.uniform(17.5, 18.5, size=(ny, nx)), ap, rng.uniform(950.0, 1020.0, size=(ny, nx)),
                                 rng.lognormal(mean=4.2, sigma=0.2, size=(ny, nx)), xcol, pw, sensor))
    out.insert(1, None)
    return out


def ssmis_granules(ctm, k: int, seed: int):
    """``k`` gridded SSMIS water-vapour granules (``satellite_ssmis``) on the model grid for ``cal_pwv``."""
    from .config import satellite_ssmis
    rng = np.random.default_rng(seed)
    lat2, lon2 = ctm[0].latitude, ctm[0].longitude
    out = []
    for g in range(k):
        vcd = rng.uniform(5.0, 60.0, size=lat2.shape)
        vcd[rng.uniform(size=vcd.shape) < 0.2] = np.nan
        vcd[1, 2] = np.inf
        out.append(satellite_ssmis(vcd, rng.uniform(0.5, 3.0, size=lat2.shape), _dt.datetime(2019, 5, 2 + g, 6, 0), lat2, lon2,
                                   False, [], 'SSMIS'))
    out.insert(1, None)
    return out
