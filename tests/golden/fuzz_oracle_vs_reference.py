#!/usr/bin/env python3
"""Differential check of the oracle against the REFERENCE ITSELF on random small cases (build container only: needs
/root/reference, imported through make_golden.py's shims; nothing here runs on the GPU box and no test imports it).

    python tests/golden/fuzz_oracle_vs_reference.py [rounds] [seed]

Where the golden fixtures pin chosen cases, this walks the corners between them: random grid ratios for ``_upscaler``, random
swaths / regions / spacings for ``interpolator`` types 1-4 (both record types), random NaN / inf / zero patterns for ``OI`` and
``error_averager``, random stacks for ``averaging``.  A mismatch is printed with the case's parameters so that it can be turned
into a fixture (that is how interpolator_rbf_ties.npz came about).  Exit code = number of mismatching cases."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_golden as mg                       # noqa: E402  (shims + reference import; its __main__ block does not run)
from oracle import oi_oracle as orc            # noqa: E402

syn, cfg = mg.syn, mg.oisat_build.config
BAD = []
CHECKS = [0]


def close(a, b, tol, what, params):
    CHECKS[0] += 1
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        BAD.append((what, params, f"shape {a.shape} vs {b.shape}"))
        return False
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        BAD.append((what, params, f"NaN pattern differs at {int((np.isnan(a) != np.isnan(b)).sum())} of {a.size}"))
        return False
    both = ~np.isnan(a)
    if both.any():
        with np.errstate(invalid="ignore"):
            same_inf = np.array_equal(np.isinf(a[both]), np.isinf(b[both]))
            fin = np.isfinite(a[both]) & np.isfinite(b[both])
            scale = np.abs(b[both][fin]).max() if fin.any() else 1.0
            err = np.abs(a[both][fin] - b[both][fin]).max() if fin.any() else 0.0
        if not same_inf or err > tol * max(scale, 1e-300):
            BAD.append((what, params, f"max abs diff {err:.3e} at scale {scale:.3e}"))
            return False
    return True


def sprinkle(rng, a, frac, values):
    a = a.copy()
    for v in values:
        a[rng.uniform(size=a.shape) < frac] = v
    return a


def fuzz_oi(rng):
    ny, nx = int(rng.integers(1, 24)), int(rng.integers(1, 24))
    Xa = rng.uniform(0.1, 10.0, (ny, nx))
    Y = sprinkle(rng, rng.uniform(-1.0, 12.0, (ny, nx)), 0.1, [np.nan])
    Sa = sprinkle(rng, rng.uniform(0.01, 9.0, (ny, nx)), 0.05, [0.0, np.nan])
    So = sprinkle(rng, rng.uniform(0.01, 9.0, (ny, nx)), 0.05, [np.inf, np.nan, 0.0])
    for on, idx in ((False, None), (True, None), (True, int(rng.integers(0, 99)))):
        mg.FORCED["knee_index"] = idx
        Yr, Yo = Y.copy(), Y.copy()
        with np.errstate(all="ignore"):
            ref = mg.quiet(mg.REF_OI, Xa.copy(), Yr, Sa.copy(), So.copy(), regularization_on=on)
            got = orc.OI(Xa.copy(), Yo, Sa.copy(), So.copy(), regularization_on=on, forced_index=(idx if on else None) if idx is not None else (0 if on else None))
        p = dict(ny=ny, nx=nx, on=on, idx=idx)
        for k in range(4):
            close(got[k], ref[k], 1e-12, f"OI out {k}", p)
        close(Yo, Yr, 0.0, "OI clamps Y in place", p)


def fuzz_error_averager(rng):
    k, ny, nx = int(rng.integers(1, 9)), int(rng.integers(1, 7)), int(rng.integers(1, 7))
    e = sprinkle(rng, rng.uniform(0.0, 4.0, (k, ny, nx)), 0.2, [np.nan, np.inf])
    with np.errstate(all="ignore"):
        close(orc.error_averager(e.copy()), mg.REF_error_averager(e.copy()), 1e-13, "error_averager", dict(k=k, ny=ny, nx=nx))


class _Reader:
    pass


def fuzz_averaging(rng):
    ny, nx, k, seed = int(rng.integers(4, 20)), int(rng.integers(4, 20)), int(rng.integers(1, 9)), int(rng.integers(1, 10 ** 6))
    stack = syn.granule_stack(ny, nx, k, seed)
    r1, r2 = _Reader(), _Reader()
    r1.sat_data = [None if g is None else mg.to_ref(g) for g in stack]
    r2.sat_data = stack
    with np.errstate(all="ignore"):
        ref = mg.quiet(mg.REF_averaging, "2019-06-01", "2019-07-01", r1)
        got = mg.quiet(orc.averaging, "2019-06-01", "2019-07-01", r2, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
    p = dict(ny=ny, nx=nx, k=k, seed=seed)
    for a, b, nm in zip(got[:5], ref[:5], ("sat_vcd", "sat_err", "ctm_vcd", "aux1", "aux2")):
        close(a, b, 1e-13, "averaging " + nm, p)
    if abs(got[5].timestamp() - ref[5].timestamp()) > 1e-3:
        BAD.append(("averaging time", p, f"{got[5]} vs {ref[5]}"))


def fuzz_upscaler(rng):
    gs = float(rng.choice([0.1, 0.25, 0.5, 1.0]))
    lat0, lon0 = float(rng.integers(-60, 40)), float(rng.integers(-150, 120))
    nlat, nlon = int(rng.integers(6, 40)), int(rng.integers(6, 40))
    lon = np.arange(lon0, lon0 + nlon * gs + gs, gs)
    lat = np.arange(lat0, lat0 + nlat * gs + gs, gs)
    X, Y = np.meshgrid(lon, lat)
    Z = sprinkle(rng, 1.0 + np.sin(X / 3.0) * np.cos(Y / 5.0) + 0.01 * rng.normal(size=X.shape), 0.03, [np.nan])
    dlat = gs * float(rng.choice([0.8, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0, 7.0]))
    dlon = gs * float(rng.choice([0.8, 1.0, 1.25, 2.0, 2.5, 5.0, 6.0]))
    ctm = syn.regional_ctm_grid(lat0, lat0 + nlat * gs, lon0, lon0 + nlon * gs, dlat, dlon)
    if ctm["Latitude"].shape[0] < 2 or ctm["Latitude"].shape[1] < 2:
        return
    thr = np.sqrt(dlat ** 2 + dlon ** 2)
    for err in (False, True):
        ref = mg.REF_interp._upscaler(X, Y, Z.copy(), ctm, gs, thr, error=err)
        got = orc.upscaler(X, Y, Z.copy(), ctm, gs, thr, error=err)
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
    nscan, npix = int(rng.integers(20, 70)), int(rng.integers(8, 30))
    if kind == "amf":
        g = syn.swath_granule(seed, nscan=nscan, npix=npix, lat0=lat0, lat1=lat1, lon_c=lon_c, width_deg=width)
        rec = cfg.satellite_amf
    elif kind in ("lattice", "noisy_lattice"):       # a level-3 record; with 1e-13 deg of noise qhull closes the hull with slivers
        step = float(rng.choice([0.5, 1.0]))
        lat0, lat1 = np.floor(lat0) + 0.5, np.floor(lat0) + 0.5 + step * int(rng.integers(8, 24))
        lon0 = np.floor(lon_c) + 0.5
        lon1 = lon0 + step * int(rng.integers(8, 24))
        g = syn.lattice_l3_granule(seed, sensor=str(rng.choice(["MOPITT", "GOSAT"])), nz=int(rng.integers(2, 4)), lat0=lat0, lat1=lat1, lon0=lon0, lon1=lon1, step=step)
        rec = cfg.satellite_opt
        if kind == "noisy_lattice":
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
        with np.errstate(all="ignore"):
            try:
                ref = mg.quiet(mg.REF_interp.interpolator, it, gs, mg.to_ref(g), ctm, thresh)
                ref_exc = None
            except Exception as e:                       # noqa: BLE001
                ref, ref_exc = None, type(e).__name__
            try:
                got = mg.quiet(orc.interpolator, it, gs, g, ctm, thresh, record_type=rec)
                got_exc = None
            except Exception as e:                       # noqa: BLE001
                got, got_exc = None, type(e).__name__
        if ref_exc != got_exc:
            BAD.append(("interpolator exception", p, f"{got_exc} vs {ref_exc}"))
            continue
        if (ref is None) != (got is None):
            BAD.append(("interpolator None", p, f"{got is None} vs {ref is None}"))
            continue
        if ref is None:
            continue
        import dataclasses
        for f in dataclasses.fields(ref):
            a, b = getattr(got, f.name), getattr(ref, f.name)
            if isinstance(b, np.ndarray) and b.shape != (1,):
                close(a, b, 1e-9 if it == 3 else 1e-11, f"interpolator type {it} {f.name}", p)
        if bool(got.ctm_upscaled_needed) != bool(ref.ctm_upscaled_needed):
            BAD.append(("interpolator need flag", p, ""))


def fuzz_amf_recal(rng):
    import copy
    ny, nx, nz = int(rng.integers(5, 16)), int(rng.integers(5, 16)), int(rng.integers(4, 24))
    ndays, k, nzs = int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.integers(4, 30))
    averaged, with_sw, with_trop = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    s1, s2 = int(rng.integers(1, 10 ** 6)), int(rng.integers(1, 10 ** 6))
    ctm = syn.ctm_days(ny, nx, nz, ndays, s1, averaged=averaged, dtype=np.float64 if rng.integers(0, 2) else np.float32)
    sat = syn.amf_granules(ctm, nzs, k, s2, with_sw=with_sw, with_trop=with_trop)
    p = dict(ny=ny, nx=nx, nz=nz, ndays=ndays, k=k, nzs=nzs, averaged=averaged, with_sw=with_sw, with_trop=with_trop, s1=s1, s2=s2)
    with np.errstate(all="ignore"):
        ref = mg.quiet(mg.REF_amf_recal, [mg.to_ref(c) for c in ctm], [mg.to_ref(x) for x in copy.deepcopy(sat)])
        got = mg.quiet(orc.amf_recal, ctm, copy.deepcopy(sat))
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
            close(getattr(a, f), getattr(b, f), 1e-11, "amf_recal " + f, p)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
    for name, fn, mult in (("OI", fuzz_oi, 3), ("error_averager", fuzz_error_averager, 5), ("averaging", fuzz_averaging, 1), ("upscaler", fuzz_upscaler, 2),
                           ("interpolator", fuzz_interpolator, 1), ("amf_recal", fuzz_amf_recal, 2)):
        before, checks = len(BAD), CHECKS[0]
        for _ in range(rounds * mult):
            fn(rng)
        print(f"{name}: {rounds * mult} cases, {CHECKS[0] - checks} arrays compared, {len(BAD) - before} mismatching", flush=True)
    for what, params, msg in BAD[:40]:
        print("MISMATCH", what, msg, params)
    return min(len(BAD), 255)


if __name__ == "__main__":
    sys.exit(main())
