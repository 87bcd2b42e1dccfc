#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN hot-path
functions on seeded synthetic inputs.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

The reference's files are imported from where they lie (never copied).  Import shims
 -- ordinary Python import errors, not refusals:
  1. a bare package object for ``oisatgmi`` so that ``oisatgmi/__init__.py`` (-> driver -> reader
     -> netCDF4, absent) is not executed;
  2. ``scipy.interpolate.interpnd._ndim_coords_from_arrays`` aliased to its scipy-1.15 home;
  3. a stub ``kneed`` module whose ``KneeLocator`` records (x, y) and returns a FORCED knee --
     the real package is not installed, so the knee pick itself stays unpinned.
Only data (inputs, outputs) is written; no reference source text is stored.
"""
import dataclasses
import importlib.util
import os
import sys
import types
import io
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

# ---- our synthetic generators, loaded under a private name (the name 'oisatgmi' is taken below)
spec = importlib.util.spec_from_file_location(
    "oisat_build", os.path.join(ROOT, "oi-sat-gmi_amd", "oisatgmi", "__init__.py"),
    submodule_search_locations=[os.path.join(ROOT, "oi-sat-gmi_amd", "oisatgmi")])
oisat_build = importlib.util.module_from_spec(spec)
sys.modules["oisat_build"] = oisat_build
spec.loader.exec_module(oisat_build)
import oisat_build.synthetic as syn          # noqa: E402

# ---- shims + reference import
pkg = types.ModuleType("oisatgmi")
pkg.__path__ = [os.path.join(REF, "oisatgmi")]
sys.modules["oisatgmi"] = pkg

import scipy.interpolate._interpnd as _ip    # noqa: E402
old = types.ModuleType("scipy.interpolate.interpnd")
old._ndim_coords_from_arrays = _ip._ndim_coords_from_arrays
sys.modules["scipy.interpolate.interpnd"] = old

FORCED = {"knee_index": None, "seen": None}
kneed = types.ModuleType("kneed")


class KneeLocator:                                    # recording stub
    def __init__(self, x, y, **kw):
        FORCED["seen"] = (np.array(x, dtype=np.float64), np.array(y, dtype=np.float64), dict(kw))
        i = FORCED["knee_index"]
        self.knee = None if i is None else x[i]


kneed.KneeLocator = KneeLocator
sys.modules["kneed"] = kneed

sys.dont_write_bytecode = True
from oisatgmi.optimal_interpolation import OI as REF_OI                 # noqa: E402
from oisatgmi.averaging import averaging as REF_averaging, error_averager as REF_error_averager  # noqa: E402
from oisatgmi import interpolator as REF_interp                         # noqa: E402
from oisatgmi import config as REF_cfg                                  # noqa: E402
from oisatgmi.amf_recal import amf_recal as REF_amf_recal               # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def to_ref(rec):
    """our record -> the reference's record class, positionally"""
    if rec is None:
        return None
    cls = getattr(REF_cfg, type(rec).__name__)
    return cls(*[getattr(rec, f.name) for f in dataclasses.fields(rec)])


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}  ({os.path.getsize(path)/1024:.0f} KiB)")


FORCED_IDX = (0, 7, 37, 98)


def gen_oi(tag, ny, nx, nobs, seed, full, extra=None):
    """OI(reg=False), the 99-point curve, OI(reg=True) at forced indices."""
    extra = extra or {}
    c = syn.diag_case(ny, nx, nobs, seed, **extra)
    # a few structural specials: Sa == 0 cell (AK NaN, K = 0), inf obs error, NaN background
    Xa, Y, Sa, So = c.Xa.copy(), c.Y.copy(), c.Sa.copy(), c.So.copy()
    obs = np.argwhere(~np.isnan(Y))
    (i0, j0), (i1, j1), (i2, j2) = obs[0], obs[1], obs[2]
    Sa[i0, j0] = 0.0
    So[i1, j1] = np.inf
    Xa[i2, j2] = np.nan
    Sa[i2, j2] = np.nan
    out = {"ny": ny, "nx": nx, "nobs": nobs, "seed": seed}
    if full:
        out.update(Xa=Xa, Y=Y, Sa=Sa, So=So)
    else:
        out.update(special=np.array([[i0, j0], [i1, j1], [i2, j2]]))
    stride = 1 if full else 97

    def pack(prefix, res, Yafter):
        for nm, a in zip(("Xb", "AK", "inc", "err"), res):
            out[f"{prefix}_{nm}"] = a.ravel()[::stride].copy()
            out[f"{prefix}_{nm}_nansum"] = np.nansum(a)
            out[f"{prefix}_{nm}_nnan"] = int(np.isnan(a).sum())
        out[f"{prefix}_Yafter_nneg"] = int((Yafter < 0).sum())

    Yw = Y.copy()
    res = quiet(REF_OI, Xa.copy(), Yw, Sa.copy(), So.copy(), regularization_on=False)
    pack("off", res, Yw)
    out["Y_clamped"] = Yw.ravel()[::stride].copy()
    for fi in FORCED_IDX:
        FORCED["knee_index"] = fi
        Yw = Y.copy()
        res = quiet(REF_OI, Xa.copy(), Yw, Sa.copy(), So.copy(), regularization_on=True)
        pack(f"on{fi}", res, Yw)
    x, y, kw = FORCED["seen"]
    out["curve_x"] = x
    out["curve_y"] = y
    out["kneed_kwargs"] = np.array(sorted(f"{k}={v}" for k, v in kw.items()))
    # knee None -> fallback index 0 (optimal_interpolation.py:40-41)
    FORCED["knee_index"] = None
    Yw = Y.copy()
    res = quiet(REF_OI, Xa.copy(), Yw, Sa.copy(), So.copy(), regularization_on=True)
    pack("onNone", res, Yw)
    out["stride"] = stride
    save(f"oi_{tag}.npz", **out)


def gen_error_averager():
    rng = np.random.default_rng(4242)
    e = rng.uniform(0.01, 1.0, size=(20, 6, 12))
    e[rng.uniform(size=e.shape) < 0.4] = np.nan
    e[:, 0, 0] = np.nan                      # all-NaN cell -> 0/0 -> NaN
    e[3, 1, 1] = np.inf                      # inf is dropped
    e[:, 2, 2] = np.inf                      # all-inf cell
    e[:, 3, 3] = np.nan
    e[5, 3, 3] = 0.25                        # single valid
    out = REF_error_averager(e.copy())
    save("error_averager.npz", inp=e, out=out)


def gen_averaging():
    for tag, (ny, nx, k, seed) in {"72x144_k5": (72, 144, 5, 2005), "36x72_k9": (36, 72, 9, 2009)}.items():
        stack = syn.granule_stack(ny, nx, k, seed)

        class R:
            pass
        r = R()
        r.sat_data = [to_ref(g) for g in stack]
        res = quiet(REF_averaging, "2019-06-01", "2019-07-01", r)
        save(f"averaging_{tag}.npz", ny=ny, nx=nx, k=k, seed=seed,
             sat_vcd=res[0], sat_err=res[1], ctm_vcd=res[2], aux1=res[3], aux2=res[4],
             avg_ts=np.float64(res[5].timestamp()))


def gen_upscaler():
    rng = np.random.default_rng(3003)
    gs = 0.25
    lon = np.arange(-10.0, 10.0 + gs, gs)
    lat = np.arange(30.0, 45.0 + gs, gs)
    X, Y = np.meshgrid(lon, lat)
    Z = 1.0 + np.sin(X / 3.0) * np.cos(Y / 5.0) + 0.01 * rng.normal(size=X.shape)
    Z[rng.uniform(size=Z.shape) < 0.01] = np.nan
    Z[0:3, 0:5] = np.nan
    out = {"X": X, "Y": Y, "Z": Z, "grid_size": gs}
    cases = {"1x1": (0.25, 0.25), "10x10": (2.5, 2.5), "8x10": (2.0, 2.5), "pass": (0.2, 0.2)}
    for tag, (dlat, dlon) in cases.items():
        ctm = syn.regional_ctm_grid(30.0, 45.0, -10.0, 10.0, dlat, dlon)
        thr = np.sqrt(dlat ** 2 + dlon ** 2)
        for err in (False, True):
            ox, oy, oz, need = REF_interp._upscaler(X, Y, Z.copy(), ctm, gs, thr, error=err)
            k = f"{tag}_{'var' if err else 'mean'}"
            out[k + "_Z"] = oz
            out[k + "_need"] = need
            out[k + "_clat"] = ctm["Latitude"]
            out[k + "_clon"] = ctm["Longitude"]
    # kernel tables (interpolator.py:40-46)
    out["box_3_4"] = REF_interp._boxfilter(3, 4)
    out["box2_3_4"] = REF_interp._boxfilter2(3, 4)
    save("upscaler.npz", **out)


def gen_interpolator():
    out = {}
    g = syn.swath_granule(5005)
    for f in ("vcd", "amf", "uncertainty", "quality_flag", "latitude_center", "longitude_center"):
        out["in_" + f] = getattr(g, f)
    for tag, (dlat, dlon, gs) in {"fine": (0.25, 0.25, 0.25), "coarse": (2.0, 2.5, 0.25)}.items():
        ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, dlat, dlon)
        out[f"{tag}_clat"] = ctm["Latitude"]
        out[f"{tag}_clon"] = ctm["Longitude"]
        out[f"{tag}_gs"] = gs
        for it in (4, 2, 1):
            r = quiet(REF_interp.interpolator, it, gs, to_ref(g), ctm, 0.75)
            assert r is not None
            for f in ("vcd", "amf", "uncertainty", "latitude_center", "longitude_center"):
                out[f"{tag}_t{it}_{f}"] = np.asarray(getattr(r, f))
            out[f"{tag}_t{it}_need"] = r.ctm_upscaled_needed
    # a granule that misses the region -> None (interpolator.py:165-167)
    ctm = syn.regional_ctm_grid(-80.0, -60.0, 100.0, 140.0, 2.0, 2.5)
    r = quiet(REF_interp.interpolator, 4, 0.25, to_ref(g), ctm, 0.75)
    out["miss_is_none"] = r is None
    save("interpolator.npz", **out)


def gen_interpolator_rbf():
    """interpolator type 3 (RBFInterpolator, 5 neighbours, thin-plate spline; interpolator.py:21-27), kept in
    its own file: the reference solves one 8x8 system per distinct neighbourhood in a Python loop."""
    out = {}
    g = syn.swath_granule(5005)
    for tag, (dlat, dlon, gs) in {"fine": (0.25, 0.25, 0.25), "coarse": (2.0, 2.5, 0.25)}.items():
        ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, dlat, dlon)
        r = quiet(REF_interp.interpolator, 3, gs, to_ref(g), ctm, 0.75)
        assert r is not None
        for f in ("vcd", "amf", "uncertainty", "latitude_center", "longitude_center"):
            out[f"{tag}_t3_{f}"] = np.asarray(getattr(r, f))
        out[f"{tag}_t3_need"] = r.ctm_upscaled_needed
    # _interpolosis type 3 by itself: scattered targets, a NaN value, distances handed in by the caller
    rng = np.random.default_rng(3303)
    pts = np.column_stack((g.longitude_center.ravel(), g.latitude_center.ravel()))
    Z = g.vcd.copy()
    Z[17, 23] = np.nan
    X = rng.uniform(0.0, 20.0, size=(40, 50))
    Y = rng.uniform(-10.0, 30.0, size=(40, 50))
    from scipy.spatial import cKDTree
    dists, _ = cKDTree(pts).query(np.column_stack((X.ravel(), Y.ravel())))
    dists = dists.reshape(X.shape)
    out["single_X"], out["single_Y"], out["single_Z"], out["single_dists"] = X, Y, Z, dists
    out["single_out"] = REF_interp._interpolosis(pts, Z, X, Y, 3, dists, 0.25)
    save("interpolator_rbf.npz", **out)


def gen_interpolator_rbf_ties():
    """_interpolosis type 3 on a regular lattice of points (an L3 product): targets on the lattice's symmetry lines are
    equidistant from several candidates for the fifth neighbour, and which one RBFInterpolator's ``KDTree(y).query(x, 5)``
    returns is a property of scipy's tree.  Three target sets: cell centres (four-fold ties), the lattice nodes themselves,
    and a 0.1-degree mesh that mixes tied and untied targets; all inside the lattice (beyond its edge the five neighbours
    are collinear and the call raises)."""
    from scipy.spatial import cKDTree
    gx, gy = np.meshgrid(10.0 + 0.25 * np.arange(14), -5.0 + 0.25 * np.arange(12))
    pts = np.column_stack((gx.ravel(), gy.ravel()))
    Z = np.sin(0.9 * gx) * np.cos(1.3 * gy) + 0.05 * gx
    out = {"points": pts, "Z": Z}
    sets = {
        "centres": np.meshgrid(10.125 + 0.25 * np.arange(13), -4.875 + 0.25 * np.arange(11)),
        "nodes": np.meshgrid(10.0 + 0.25 * np.arange(14), -5.0 + 0.25 * np.arange(12)),
        "mesh": np.meshgrid(10.05 + 0.1 * np.arange(32), -4.95 + 0.1 * np.arange(27)),
    }
    for tag, (X, Y) in sets.items():
        dists, _ = cKDTree(pts).query(np.column_stack((X.ravel(), Y.ravel())))
        dists = dists.reshape(X.shape)
        out[f"{tag}_X"], out[f"{tag}_Y"], out[f"{tag}_dists"] = X, Y, dists
        out[f"{tag}_out"] = REF_interp._interpolosis(pts, Z, X, Y, 3, dists, 0.25)
    # interpolator(3, ...) on a level-3 lattice record (per-level cubes: several field stacks over one set of neighbourhoods)
    for sensor, seed, grid in (("MOPITT", 6201, "gs100_1x125"), ("GOSAT", 6202, "gs100_2x25")):
        la0, la1, lo0, lo1, dlat, dlon, gs = TIE_GRIDS[grid]
        ctm = syn.regional_ctm_grid(la0, la1, lo0, lo1, dlat, dlon)
        out[f"{grid}_clat"], out[f"{grid}_clon"] = ctm["Latitude"], ctm["Longitude"]
        r = quiet(REF_interp.interpolator, 3, gs, to_ref(syn.lattice_l3_granule(seed, sensor=sensor)), ctm, 0.0)
        assert r is not None
        names = []
        for f in dataclasses.fields(r):
            v = getattr(r, f.name)
            if isinstance(v, np.ndarray):
                out[f"l3_{sensor}_{grid}_t3_{f.name}"] = v
                names.append(f.name)
        out[f"l3_{sensor}_{grid}_t3_arrays"] = np.array(names)
    save("interpolator_rbf_ties.npz", **out)


def gen_interpolator_levels():
    """interpolator() on records that carry per-level cubes: the satellite_amf scattering-weight / pressure loops
    (interpolator.py:191-213) and both satellite_opt branches -- MOPITT (nz+1 averaging-kernel rows, no pressure weights)
    and GOSAT (nz rows + pressure weights) -- incl. a-priori column / surface, surface pressure, x_col, a-priori profile
    (:216-283) and the positional rebuild (:284-290).  Types 4 (k-d tree) and 1 (Delaunay), a coarse model grid (box
    filter + NN pick) and a fine one (pass-through to the 0.25 deg grid)."""
    out = {}
    grids = {"coarse": (-6.0, 26.0, 2.0, 18.0, 2.0, 2.5, 0.25), "fine": (4.0, 16.0, 6.0, 14.0, 0.25, 0.25, 0.25)}
    for kind, seed in (("amf", 6101), ("MOPITT", 6102), ("GOSAT", 6103)):
        g = syn.swath_level_granule(seed, kind=kind, nz=3)
        for f in dataclasses.fields(g):
            v = getattr(g, f.name)
            if isinstance(v, np.ndarray) and v.size > 1:
                out[f"{kind}_in_{f.name}"] = v
        for tag, (la0, la1, lo0, lo1, dlat, dlon, gs) in grids.items():
            ctm = syn.regional_ctm_grid(la0, la1, lo0, lo1, dlat, dlon)
            out[f"{tag}_clat"], out[f"{tag}_clon"], out[f"{tag}_gs"] = ctm["Latitude"], ctm["Longitude"], gs
            for it in (4, 1):
                r = quiet(REF_interp.interpolator, it, gs, to_ref(g), ctm, 0.75)
                assert r is not None
                names = []
                for f in dataclasses.fields(r):
                    v = getattr(r, f.name)
                    if isinstance(v, np.ndarray):
                        out[f"{kind}_{tag}_t{it}_{f.name}"] = v
                        names.append(f.name)
                    elif isinstance(v, (bool, str)):
                        out[f"{kind}_{tag}_t{it}_{f.name}"] = np.array(v)
                        names.append(f.name)
                out[f"{kind}_{tag}_t{it}_arrays"] = np.array(names)
    save("interpolator_levels.npz", **out)


def _fine_grid(ctm, gs):
    """the fine grid exactly as interpolator() builds it (interpolator.py:136-143)"""
    lat, lon = ctm["Latitude"], ctm["Longitude"]
    lon_grid = np.arange(np.min(lon), np.max(lon) + gs, gs)
    lat_grid = np.arange(np.min(lat), np.max(lat) + gs, gs)
    return np.meshgrid(lon_grid, lat_grid)


TIE_GRIDS = {   # tag -> (lat0, lat1, lon0, lon1, dlat, dlon, grid_size): the reference's MOPITT / GOSAT settings
    "gs100_1x125": (-20.0, 20.0, -30.0, 30.0, 1.0, 1.25, 1.0),        # grid_size 1.0 (reader.py:1209,:1271), GMI 1 x 1.25
    "gs100_2x25": (-20.0, 20.0, -30.0, 30.0, 2.0, 2.5, 1.0),          # grid_size 1.0, GMI 2 x 2.5
    "gs025_05x0625": (-10.0, 10.0, -15.0, 15.0, 0.5, 0.625, 0.25),    # 0.25 deg, MERRA2-GMI 0.5 x 0.625
    "global_1x125": (-90.0, 90.0, -180.0, 178.75, 1.0, 1.25, 1.0),    # the whole 181 x 288 model grid
}


def gen_upscaler_ties():
    """_upscaler / interpolator() where model cell centres lie EXACTLY midway between fine-grid nodes, so the
    reference's cKDTree(points).query(xi) (interpolator.py:78-91) has to pick among equidistant nodes: the model
    longitude spacing 1.25 / 2.5 deg against grid_size 1.0 (MOPITT, GOSAT), 0.625 against 0.25.  'index' fields carry
    the fine node's flat index (identifies the pick outright when the box kernel is 1x1), 'rand' fields distinct
    random values with NaN holes (identify it through the box average)."""
    out = {}
    rng = np.random.default_rng(3113)
    for tag, (la0, la1, lo0, lo1, dlat, dlon, gs) in TIE_GRIDS.items():
        ctm = syn.regional_ctm_grid(la0, la1, lo0, lo1, dlat, dlon)
        X, Y = _fine_grid(ctm, gs)
        thr = np.sqrt(dlat ** 2 + dlon ** 2)
        Zi = np.arange(X.size, dtype=np.float64).reshape(X.shape)
        Zr = rng.uniform(1.0, 2.0, size=X.shape)
        Zr[rng.uniform(size=X.shape) < 0.02] = np.nan
        out[f"{tag}_spec"] = np.array([la0, la1, lo0, lo1, dlat, dlon, gs])
        out[f"{tag}_X"], out[f"{tag}_Y"], out[f"{tag}_Zrand"] = X, Y, Zr
        out[f"{tag}_clat"], out[f"{tag}_clon"] = ctm["Latitude"], ctm["Longitude"]
        for nm, Z in (("index", Zi), ("rand", Zr)):
            for err in (False, True):
                _, _, oz, need = REF_interp._upscaler(X, Y, Z.copy(), ctm, gs, thr, error=err)
                assert need is False
                out[f"{tag}_{nm}_{'var' if err else 'mean'}"] = oz
    # interpolator() on level-3 lattice records (MOPITT MOP03 style: reader.py:1150-1211, grid_size 1.0, type 1,
    # flag_thresh 0.0), plus type 4 where the swath -> fine-grid search itself meets four-way ties
    for sensor, seed, grid in (("MOPITT", 6201, "gs100_1x125"), ("MOPITT", 6201, "gs100_2x25"), ("GOSAT", 6202, "gs100_2x25")):
        la0, la1, lo0, lo1, dlat, dlon, gs = TIE_GRIDS[grid]
        ctm = syn.regional_ctm_grid(la0, la1, lo0, lo1, dlat, dlon)
        g = syn.lattice_l3_granule(seed, sensor=sensor)
        for it in (1, 4):
            r = quiet(REF_interp.interpolator, it, gs, to_ref(g), ctm, 0.0)
            assert r is not None
            names = []
            for f in dataclasses.fields(r):
                v = getattr(r, f.name)
                if isinstance(v, np.ndarray):
                    out[f"l3_{sensor}_{grid}_t{it}_{f.name}"] = v
                    names.append(f.name)
            out[f"l3_{sensor}_{grid}_t{it}_arrays"] = np.array(names)
    save("upscaler_ties.npz", **out)


def gen_linear_degenerate():
    """_interpolosis type 1 on a triangulation that holds DEGENERATE simplices (NaN barycentric transforms): an exactly
    regular pixel lattice whose latitudes carry 1e-13 deg of noise, so qhull closes the hull with zero-area slivers.
    scipy's point location then leaves its directed walk for the brute-force scan; targets sit on and around the hull."""
    from scipy.spatial import Delaunay, cKDTree
    rng = np.random.default_rng(777)
    lon1, lat1 = np.arange(0.0, 6.0, 0.25), np.arange(10.0, 14.0, 0.25)
    LON, LAT = np.meshgrid(lon1, lat1)
    pts = np.column_stack((LON.ravel(), LAT.ravel() + 1e-13 * rng.normal(size=LON.size)))
    tri = Delaunay(pts)
    nbad = int(np.isnan(tri.transform[:, 0, 0]).sum())
    assert nbad > 0
    Z = (1.0 + np.sin(pts[:, 0]) * np.cos(pts[:, 1] / 3.0)).reshape(LON.shape)
    gx, gy = np.meshgrid(np.arange(-0.25, 6.01, 0.0625), np.arange(9.75, 14.01, 0.0625))
    # plus targets exactly on pixel rows / hull edges
    X = np.concatenate([gx.ravel(), lon1, lon1 + 0.125, np.full(lat1.size, 0.0), np.full(lat1.size, 5.75)])
    Y = np.concatenate([gy.ravel(), np.full(lon1.size, 10.0), np.full(lon1.size, 13.75), lat1 + 0.1, lat1 + 0.1])
    X, Y = X.reshape(1, -1), Y.reshape(1, -1)
    dists, _ = cKDTree(pts).query(np.column_stack((X.ravel(), Y.ravel())))
    dists = dists.reshape(X.shape)
    out = REF_interp._interpolosis(tri, Z, X, Y, 1, dists, 0.25)
    save("interpolator_degenerate.npz", pts=pts, Z=Z, X=X, Y=Y, dists=dists, out=out, n_degenerate=nbad)


def amf_cases():
    """name -> (ctm_data, sat_data) builders shared with the tests (seeded)"""
    def case_a():
        ctm = syn.ctm_days(10, 14, 12, 2, 9101, averaged=False)
        return ctm, syn.amf_granules(ctm, 9, 3, 9102, with_sw=True, with_trop=True)

    def case_b():
        ctm = syn.ctm_days(8, 9, 20, 1, 9201, averaged=True, dtype=np.float64)
        return ctm, syn.amf_granules(ctm, 35, 2, 9202, with_sw=True, with_trop=False)

    def case_c():
        ctm = syn.ctm_days(10, 14, 12, 2, 9301, averaged=False)
        return ctm, syn.amf_granules(ctm, 9, 2, 9302, with_sw=False, with_trop=True)

    def case_d():                           # model finer than the satellite grid: upscaling needed
        ctm = syn.ctm_days(49, 65, 6, 1, 9401, averaged=True, lat0=-12.0, lat1=12.0, lon0=-16.0, lon1=16.0)
        coarse = syn.ctm_days(9, 11, 6, 1, 9402, averaged=True, lat0=-10.0, lat1=10.0, lon0=-12.5, lon1=12.5)
        sat = syn.amf_granules(coarse, 7, 2, 9403, with_sw=True, with_trop=True)
        for s in sat:
            if s is not None:
                s.ctm_upscaled_needed = True
        return ctm, sat
    return {"a": case_a, "b": case_b, "c": case_c, "d": case_d}


def gen_amf_recal():
    out = {}
    for tag, build in amf_cases().items():
        ctm, sat = build()
        ref_ctm = [to_ref(c) for c in ctm]
        ref_sat = [to_ref(s) for s in sat]
        res = quiet(REF_amf_recal, ref_ctm, ref_sat)
        k = 0
        for r in res:
            if r is None:
                continue
            for f in ("vcd", "ctm_vcd", "new_amf", "old_amf"):
                out[f"{tag}_{k}_{f}"] = np.asarray(getattr(r, f), dtype=np.float64)
            out[f"{tag}_{k}_time"] = np.float64(r.ctm_time_at_sat)
            k += 1
        out[f"{tag}_n"] = k
    save("amf_recal.npz", **out)


def gen_ak_conv():
    """ak_conv_mopitt / ak_conv_gosat (the satellite_opt counterpart of amf_recal) on the seeded cases of tests/amf_cases.py"""
    from oisatgmi.ak_conv_mopitt import ak_conv_mopitt as REF_mopitt
    from oisatgmi.ak_conv_gosat import ak_conv_gosat as REF_gosat
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.modules.setdefault("oisatgmi.synthetic", syn)
    import importlib
    spec2 = importlib.util.spec_from_file_location("amf_cases_build", os.path.join(ROOT, "tests", "amf_cases.py"))
    src = open(os.path.join(ROOT, "tests", "amf_cases.py")).read().replace("from oisatgmi import synthetic as syn", "")
    ns = {"syn": syn, "np": np}
    exec(compile(src, "amf_cases.py", "exec"), ns)               # our own test helper, with our generators bound
    out = {}
    for tag, build in ns["akconv_cases"]().items():
        sensor, ctm, sat = build()
        fn = REF_mopitt if sensor == "MOPITT" else REF_gosat
        with np.errstate(all="ignore"):
            res = quiet(fn, [to_ref(c) for c in ctm], [to_ref(x) for x in sat])
        k = 0
        for r in res:
            if r is None:
                continue
            out[f"{tag}_{k}_ctm_vcd"] = np.asarray(r.ctm_vcd, dtype=np.float64)
            out[f"{tag}_{k}_ctm_xcol"] = np.asarray(r.ctm_xcol, dtype=np.float64)
            out[f"{tag}_{k}_time"] = np.float64(r.ctm_time_at_sat)
            k += 1
        out[f"{tag}_n"] = k
    save("ak_conv.npz", **out)


def gen_pwv():
    """pwv_calculator (SSMIS branch of run/job.py:69-70) on the seeded cases of tests/amf_cases.py"""
    from oisatgmi.pwv_cal import pwv_calculator as REF_pwv
    src = open(os.path.join(ROOT, "tests", "amf_cases.py")).read().replace("from oisatgmi import synthetic as syn", "")
    ns = {"syn": syn, "np": np}
    exec(compile(src, "amf_cases.py", "exec"), ns)
    out = {}
    for tag, build in ns["pwv_cases"]().items():
        ctm, sat = build()
        with np.errstate(all="ignore"):
            res = quiet(REF_pwv, [to_ref(c) for c in ctm], [to_ref(x) for x in sat])
        k = 0
        for r in res:
            if r is None:
                continue
            out[f"{tag}_{k}_ctm_vcd"] = np.asarray(r.ctm_vcd, dtype=np.float64)
            k += 1
        out[f"{tag}_n"] = k
    save("pwv.npz", **out)


def gen_records():
    out = {}
    for nm in ("satellite_amf", "satellite_opt", "satellite_ssmis", "ctm_model"):
        out[nm] = np.array([f.name for f in dataclasses.fields(getattr(REF_cfg, nm))])
    save("records.npz", **out)


if __name__ == "__main__":
    print("generating golden vectors from", REF)
    if sys.argv[1:] == ["rbf"]:                   # only the (slow) type-3 file
        gen_interpolator_rbf()
        raise SystemExit(0)
    if sys.argv[1:] == ["rbfties"]:
        gen_interpolator_rbf_ties()
        raise SystemExit(0)
    if sys.argv[1:] == ["akconv"]:
        gen_ak_conv()
        raise SystemExit(0)
    if sys.argv[1:] == ["degenerate"]:
        gen_linear_degenerate()
        raise SystemExit(0)
    if sys.argv[1:] == ["levels"]:
        gen_interpolator_levels()
        raise SystemExit(0)
    if sys.argv[1:] == ["pwv"]:
        gen_pwv()
        raise SystemExit(0)
    if sys.argv[1:] == ["ties"]:
        gen_upscaler_ties()
        raise SystemExit(0)
    gen_records()
    gen_oi("72x144", 72, 144, 1000, 1001, full=True)
    gen_oi("360x720", 360, 720, 10000, 2001, full=False)
    gen_oi("o3_72x144", 72, 144, 4000, 5003, full=True,
           extra=dict(ctm_error=10.0, value_range=(200.0, 500.0), base=250.0, amp=150.0, rel_obs_err=0.04))
    gen_error_averager()
    gen_averaging()
    gen_upscaler()
    gen_interpolator()
    gen_interpolator_rbf()
    gen_interpolator_rbf_ties()
    gen_interpolator_levels()
    gen_linear_degenerate()
    gen_upscaler_ties()
    gen_amf_recal()
    gen_ak_conv()
    gen_pwv()
    print("done")
