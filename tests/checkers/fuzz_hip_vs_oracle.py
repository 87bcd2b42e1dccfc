#!/usr/bin/env python3
"""Random small cases through the HIP path and through the oracle (the checker), side by side -- the GPU-side twin of
tests/golden/fuzz_oracle_vs_reference.py.  Run on the GPU box:

    gpurun -- python tests/checkers/fuzz_hip_vs_oracle.py [rounds] [seed]

A bug hunt, not a test: the pinned cases live in tests/; what this finds becomes a fixture or a test there."""
import dataclasses
import io
import contextlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oi-sat-gmi_amd"))
sys.path.insert(0, ROOT)
from oisatgmi import synthetic as syn, config as cfg                  # noqa: E402
from oisatgmi import interpolator as hip_interp                       # noqa: E402
from oisatgmi.optimal_interpolation import OI as HIP_OI               # noqa: E402
from oisatgmi.averaging import averaging as hip_averaging, error_averager as hip_error_averager   # noqa: E402
from oracle import oi_oracle as orc                                   # noqa: E402  (the checker)

BAD = []
CHECKS = [0]


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def close(a, b, tol, what, params):
    CHECKS[0] += 1
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        BAD.append((what, params, f"shape {a.shape} vs {b.shape}"))
        return
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        BAD.append((what, params, f"NaN pattern differs at {int((np.isnan(a) != np.isnan(b)).sum())} of {a.size}"))
        return
    ok = ~np.isnan(a)
    if ok.any():
        with np.errstate(invalid="ignore"):
            same_inf = np.array_equal(np.isinf(a[ok]), np.isinf(b[ok]))
            fin = np.isfinite(a[ok]) & np.isfinite(b[ok])
            scale = np.abs(b[ok][fin]).max() if fin.any() else 1.0
            err = np.abs(a[ok][fin] - b[ok][fin]).max() if fin.any() else 0.0
        if not same_inf or err > tol * max(scale, 1e-300):
            BAD.append((what, params, f"max abs diff {err:.3e} at scale {scale:.3e}"))


def sprinkle(rng, a, frac, values):
    a = a.copy()
    for v in values:
        a[rng.uniform(size=a.shape) < frac] = v
    return a


def fuzz_oi(rng):
    ny, nx = int(rng.integers(1, 60)), int(rng.integers(1, 60))
    Xa = rng.uniform(0.1, 10.0, (ny, nx))
    Y = sprinkle(rng, rng.uniform(-1.0, 12.0, (ny, nx)), 0.1, [np.nan])
    Sa = sprinkle(rng, rng.uniform(0.01, 9.0, (ny, nx)), 0.05, [0.0, np.nan])
    So = sprinkle(rng, rng.uniform(0.01, 9.0, (ny, nx)), 0.05, [np.inf, np.nan, 0.0])
    for on in (False, True):
        Yh, Yo = Y.copy(), Y.copy()
        with np.errstate(all="ignore"):
            got = quiet(HIP_OI, Xa.copy(), Yh, Sa.copy(), So.copy(), regularization_on=on)
            ref = quiet(orc.OI, Xa.copy(), Yo, Sa.copy(), So.copy(), regularization_on=on)
        p = dict(ny=ny, nx=nx, on=on)
        for k in range(4):
            close(got[k], ref[k], 1e-12, f"OI out {k}", p)
        close(Yh, Yo, 0.0, "OI clamps Y in place", p)


def fuzz_error_averager(rng):
    k, ny, nx = int(rng.integers(1, 12)), int(rng.integers(1, 30)), int(rng.integers(1, 30))
    e = sprinkle(rng, rng.uniform(0.0, 4.0, (k, ny, nx)), 0.2, [np.nan, np.inf])
    with np.errstate(all="ignore"):
        close(hip_error_averager(e.copy()), orc.error_averager(e.copy()), 1e-13, "error_averager", dict(k=k, ny=ny, nx=nx))


class _Reader:
    pass


def fuzz_averaging(rng):
    ny, nx, k, seed = int(rng.integers(4, 40)), int(rng.integers(4, 40)), int(rng.integers(1, 12)), int(rng.integers(1, 10 ** 6))
    r1, r2 = _Reader(), _Reader()
    r1.sat_data = syn.granule_stack(ny, nx, k, seed)
    r2.sat_data = syn.granule_stack(ny, nx, k, seed)
    with np.errstate(all="ignore"):
        got = quiet(hip_averaging, "2019-06-01", "2019-07-01", r1)
        ref = quiet(orc.averaging, "2019-06-01", "2019-07-01", r2, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
    p = dict(ny=ny, nx=nx, k=k, seed=seed)
    for a, b, nm in zip(got[:5], ref[:5], ("sat_vcd", "sat_err", "ctm_vcd", "aux1", "aux2")):
        close(a, b, 1e-13, "averaging " + nm, p)


def fuzz_upscaler(rng):
    gs = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
    lat0, lon0 = float(rng.integers(-60, 40)), float(rng.integers(-150, 120))
    nlat, nlon = int(rng.integers(6, 60)), int(rng.integers(6, 60))
    lon = np.arange(lon0, lon0 + nlon * gs + gs, gs)
    lat = np.arange(lat0, lat0 + nlat * gs + gs, gs)
    X, Y = np.meshgrid(lon, lat)
    Z = sprinkle(rng, 1.0 + np.sin(X / 3.0) * np.cos(Y / 5.0) + 0.01 * rng.normal(size=X.shape), 0.03, [np.nan])
    dlat = gs * float(rng.choice([0.8, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 7.0]))
    dlon = gs * float(rng.choice([0.8, 1.0, 1.25, 2.0, 2.5, 5.0, 6.0]))
    ctm = syn.regional_ctm_grid(lat0, lat0 + nlat * gs, lon0, lon0 + nlon * gs, dlat, dlon)
    if min(ctm["Latitude"].shape) < 2:
        return
    thr = np.sqrt(dlat ** 2 + dlon ** 2)
    for err in (False, True):
        got = hip_interp._upscaler(X, Y, Z.copy(), ctm, gs, thr, error=err)
        ref = orc.upscaler(X, Y, Z.copy(), ctm, gs, thr, error=err)
        p = dict(gs=gs, lat0=lat0, lon0=lon0, nlat=nlat, nlon=nlon, dlat=dlat, dlon=dlon, err=err)
        if bool(ref[3]) != bool(got[3]):
            BAD.append(("upscaler need flag", p, f"{got[3]} vs {ref[3]}"))
        else:
            close(got[2], ref[2], 1e-12, "upscaler Z", p)


def fuzz_interpolator(rng):
    kind = str(rng.choice(["amf", "amf_levels", "MOPITT", "GOSAT", "lattice", "noisy_lattice"]))
    seed = int(rng.integers(1, 10 ** 6))
    lat0 = float(rng.uniform(-40, 10))
    lat1 = lat0 + float(rng.uniform(8, 30))
    lon_c = float(rng.uniform(-60, 60))
    width = float(rng.uniform(4, 16))
    nscan, npix = int(rng.integers(20, 90)), int(rng.integers(8, 40))
    if kind == "amf":
        g = syn.swath_granule(seed, nscan=nscan, npix=npix, lat0=lat0, lat1=lat1, lon_c=lon_c, width_deg=width)
        rec = cfg.satellite_amf
    elif kind in ("lattice", "noisy_lattice"):
        step = float(rng.choice([0.5, 1.0]))
        lat0, lat1 = np.floor(lat0) + 0.5, np.floor(lat0) + 0.5 + step * int(rng.integers(8, 24))
        lon0 = np.floor(lon_c) + 0.5
        lon1 = lon0 + step * int(rng.integers(8, 24))
        g = syn.lattice_l3_granule(seed, sensor=str(rng.choice(["MOPITT", "GOSAT"])), nz=int(rng.integers(2, 4)), lat0=lat0, lat1=lat1, lon0=lon0, lon1=lon1, step=step)
        rec = cfg.satellite_opt
        if kind == "noisy_lattice":      # 1e-13 deg of noise on the latitudes: qhull closes the hull with zero-area slivers (NaN transforms)
            g.latitude_center = g.latitude_center.astype(np.float64) + 1e-13 * rng.normal(size=g.latitude_center.shape)
            g.longitude_center = g.longitude_center.astype(np.float64)
        lon_c, width = 0.5 * (lon0 + lon1), lon1 - lon0
    else:
        g = syn.swath_level_granule(seed, kind="amf" if kind == "amf_levels" else kind, nz=int(rng.integers(2, 5)), nscan=nscan, npix=npix, lat0=lat0,
                                    lat1=lat1, lon_c=lon_c, width_deg=width)
        rec = cfg.satellite_amf if kind == "amf_levels" else cfg.satellite_opt
    gs = float(rng.choice([0.25, 0.5, 1.0]))
    dlat = gs * float(rng.choice([0.5, 1.0, 2.0, 4.0]))
    dlon = gs * float(rng.choice([0.5, 1.0, 2.5, 5.0]))
    pad = float(rng.uniform(-3, 3)) if "lattice" not in kind else float(rng.uniform(-3, -0.5))    # beyond a lattice's edge type 3 is singular
    ctm = syn.regional_ctm_grid(np.floor(lat0 - pad), np.ceil(lat1 + pad), np.floor(lon_c - width / 2 - pad), np.ceil(lon_c + width / 2 + pad), dlat, dlon)
    if min(ctm["Latitude"].shape) < 2:
        return
    thresh = float(rng.choice([0.0, 0.5, 0.75]))
    for it in (4, 2, 1, 3):
        p = dict(kind=kind, seed=seed, lat0=lat0, lat1=lat1, lon_c=lon_c, width=width, nscan=nscan, npix=npix, gs=gs, dlat=dlat, dlon=dlon, pad=pad, thresh=thresh, it=it)
        res = []
        for fn, kw in ((hip_interp.interpolator, {}), (orc.interpolator, {"record_type": rec})):
            with np.errstate(all="ignore"):
                try:
                    res.append((quiet(fn, it, gs, g, ctm, thresh, **kw), None))
                except Exception as e:                       # noqa: BLE001
                    res.append((None, type(e).__name__))
        (got, got_exc), (ref, ref_exc) = res
        if ref_exc != got_exc:
            BAD.append(("interpolator exception", p, f"{got_exc} vs {ref_exc}"))
            continue
        if (ref is None) != (got is None):
            BAD.append(("interpolator None", p, f"{got is None} vs {ref is None}"))
            continue
        if ref is None:
            continue
        for f in dataclasses.fields(ref):
            a, b = getattr(got, f.name), getattr(ref, f.name)
            if isinstance(b, np.ndarray) and b.shape != (1,):
                close(a, b, 1e-9 if it == 3 else 1e-11, f"interpolator type {it} {f.name}", p)
        if bool(got.ctm_upscaled_needed) != bool(ref.ctm_upscaled_needed):
            BAD.append(("interpolator need flag", p, ""))


def fuzz_amf_recal(rng):
    import copy
    from oisatgmi.amf_recal import amf_recal as hip_amf_recal
    ny, nx, nz = int(rng.integers(5, 40)), int(rng.integers(5, 40)), int(rng.integers(4, 24))
    ndays, k, nzs = int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.integers(4, 30))
    averaged, with_sw, with_trop = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    s1, s2 = int(rng.integers(1, 10 ** 6)), int(rng.integers(1, 10 ** 6))
    f64 = bool(rng.integers(0, 2))
    ctm = syn.ctm_days(ny, nx, nz, ndays, s1, averaged=averaged, dtype=np.float64 if f64 else np.float32)
    sat = syn.amf_granules(ctm, nzs, k, s2, with_sw=with_sw, with_trop=with_trop)
    p = dict(ny=ny, nx=nx, nz=nz, ndays=ndays, k=k, nzs=nzs, averaged=averaged, with_sw=with_sw, with_trop=with_trop, s1=s1, s2=s2, f64=f64)
    with np.errstate(all="ignore"):
        got = quiet(hip_amf_recal, ctm, copy.deepcopy(sat))
        ref = quiet(orc.amf_recal, ctm, copy.deepcopy(sat))
    if len(ref) != len(got):
        BAD.append(("amf_recal length", p, f"{len(got)} vs {len(ref)}"))
        return
    for a, b in zip(got, ref):
        if (a is None) != (b is None):
            BAD.append(("amf_recal None", p, ""))
            continue
        if b is None:
            continue
        for f in ("vcd", "ctm_vcd", "new_amf", "old_amf"):
            if np.shape(getattr(b, f)) == (1,):           # np.empty((1)) placeholder (amf_recal.py:169-170): uninitialised
                if np.shape(getattr(a, f)) != (1,):
                    BAD.append(("amf_recal placeholder", p, f))
                continue
            close(getattr(a, f), getattr(b, f), 1e-11 if f64 else 1e-5, "amf_recal " + f, p)


def fuzz_dense(rng):
    """The dense Gaussian-B analysis (no reference code: the oracle is the float64 restatement) on random small months."""
    from oisatgmi import dense, _hip
    ny, nx = [(18, 36), (36, 72), (45, 90), (72, 144)][int(rng.integers(0, 4))]
    m = int(rng.choice([0, 1, 3, 60, 127, 128, 129, 300, 700, 1500, 2100, 4200, 6100]))     # (the larger ones reach the task graph)
    L = float(np.exp(rng.uniform(np.log(80.0), np.log(3000.0))))
    species = str(rng.choice(["NO2", "HCHO", "O3"]))
    seed = int(rng.integers(1, 10 ** 6))
    swaths = bool(rng.integers(0, 2)) and m >= 60
    p = syn.point_obs_case(ny, nx, max(m, 1), seed, swaths=swaths, species=species)
    olat, olon, y, var = p.obs_lat[:m].copy(), p.obs_lon[:m].copy(), p.obs_y[:m].copy(), p.obs_var[:m].copy()
    m = olat.size
    how = str(rng.choice(["plain", "duplicates", "cluster", "small_R", "large_R"]))
    if m >= 3 and how == "duplicates":
        k = max(1, m // 4)
        olat[:k], olon[:k] = olat[-k:], olon[-k:]
    elif m >= 3 and how == "cluster":
        olat = 10.0 + 3.0 * rng.normal(size=m)
        olon = 20.0 + 3.0 * rng.normal(size=m)
    elif how == "small_R":
        var = var * 0.05
    elif how == "large_R":
        var = var * 50.0
    dt = np.float64 if rng.integers(0, 2) else np.float32
    prm = dict(ny=ny, nx=nx, m=m, L=round(L, 1), species=species, seed=seed, swaths=swaths, how=how, dt=np.dtype(dt).name)
    cell = dense.regular_grid_cell(p.lat, p.lon, olat, olon)
    try:
        xb, inc, info = dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, L, refine=3, dtype=dt, obs=dict(lat=olat, lon=olon, y=y, var=var))
    except _hip.OisatError as e:
        # an honest refusal (not positive definite in float32 / refinement above tolerance) is an answer; a wrong field is not
        msg = str(e)
        if "refinement tolerance" in msg or "positive definite" in msg:
            prm["refused"] = msg[:60]
            REFUSED.append(prm)
            return
        BAD.append(("dense raised", prm, msg[:200]))
        return
    ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, olat, olon, cell, np.where(y < 0, 0, y), var, L)
    tol = 1e-5                                   # north_star's bound (the refinement stops at a 1e-6 residual: ill-conditioned
    scale = np.abs(ref["xa"]).max()              # months -- tiny R, clustered observations, long L -- reach 4e-6 of the field scale)
    CHECKS[0] += 2
    for nm, a, b in (("xa", xb.ravel(), ref["xa"]), ("inc", inc.ravel(), ref["inc"])):
        err = np.abs(a - b).max() if a.size else 0.0
        WORST[0] = max(WORST[0], err / scale)
        if not np.array_equal(np.isnan(a), np.isnan(b)) or err > tol * scale:
            BAD.append(("dense " + nm, prm, f"max abs diff {err:.3e} = {err / scale:.2e} of scale"))


REFUSED = []
WORST = [0.0]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
    for name, fn, mult in (("OI", fuzz_oi, 3), ("error_averager", fuzz_error_averager, 5), ("averaging", fuzz_averaging, 1), ("upscaler", fuzz_upscaler, 2),
                           ("interpolator", fuzz_interpolator, 1), ("amf_recal", fuzz_amf_recal, 2), ("dense", fuzz_dense, 2)):
        before, checks = len(BAD), CHECKS[0]
        for _ in range(rounds * mult):
            fn(rng)
        print(f"{name}: {rounds * mult} cases, {CHECKS[0] - checks} arrays compared, {len(BAD) - before} mismatching", flush=True)
    print(f"dense: worst field error {WORST[0]:.2e} of the field scale (bound 1e-5), {len(REFUSED)} refusals")
    for prm in REFUSED:
        print("refused (counted as an answer):", prm)
    for what, params, msg in BAD[:60]:
        print("MISMATCH", what, msg, params)
    return min(len(BAD), 255)


if __name__ == "__main__":
    sys.exit(main())
