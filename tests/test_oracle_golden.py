"""The CPU oracle (oracle/oi_oracle.py) against the golden vectors produced by the reference's
own functions (tests/golden/make_golden.py).  CPU only.  Tolerances: the oracle restates the
same float64 arithmetic, so 1e-12 relative unless a summation order differs (noted)."""
import dataclasses
import datetime

import numpy as np
import pytest

from oracle import oi_oracle as orc
from oisatgmi import synthetic as syn
from oisatgmi import config as cfg

FORCED_IDX = (0, 7, 37, 98)


def _specials(c, g):
    Xa, Y, Sa, So = c.Xa.copy(), c.Y.copy(), c.Sa.copy(), c.So.copy()
    (i0, j0), (i1, j1), (i2, j2) = g["special"]
    Sa[i0, j0] = 0.0
    So[i1, j1] = np.inf
    Xa[i2, j2] = np.nan
    Sa[i2, j2] = np.nan
    return Xa, Y, Sa, So


def _oi_inputs(g, tag):
    if "Xa" in g.files:
        return g["Xa"].copy(), g["Y"].copy(), g["Sa"].copy(), g["So"].copy()
    c = syn.diag_case(int(g["ny"]), int(g["nx"]), int(g["nobs"]), int(g["seed"]))
    return _specials(c, g)


def _check_pack(g, prefix, res, Yafter, rtol=1e-12):
    stride = int(g["stride"])
    for nm, a in zip(("Xb", "AK", "inc", "err"), res):
        np.testing.assert_allclose(a.ravel()[::stride], g[f"{prefix}_{nm}"], rtol=rtol, atol=0, equal_nan=True)
        assert int(np.isnan(a).sum()) == int(g[f"{prefix}_{nm}_nnan"])
        np.testing.assert_allclose(np.nansum(a), g[f"{prefix}_{nm}_nansum"], rtol=1e-10)
    assert int((Yafter < 0).sum()) == 0 == int(g[f"{prefix}_Yafter_nneg"])


@pytest.mark.parametrize("tag", ["72x144", "360x720", "o3_72x144"])
def test_oi_against_reference(golden, tag):
    g = golden(f"oi_{tag}.npz")
    Xa, Y, Sa, So = _oi_inputs(g, tag)
    Yw = Y.copy()
    res = orc.OI(Xa.copy(), Yw, Sa, So, regularization_on=False)
    _check_pack(g, "off", res[:4], Yw)
    np.testing.assert_array_equal(Yw.ravel()[::int(g["stride"])], g["Y_clamped"])
    for fi in FORCED_IDX:
        Yw = Y.copy()
        res = orc.OI(Xa.copy(), Yw, Sa, So, regularization_on=True, forced_index=fi)
        _check_pack(g, f"on{fi}", res[:4], Yw)
    # the 99-point curve is pinned by the reference; nanmean summation order is numpy's in both
    np.testing.assert_allclose(np.array(orc.scaling_factors(True)), g["curve_x"], rtol=0, atol=0)
    np.testing.assert_allclose(res[4], g["curve_y"], rtol=1e-13)
    Yw = Y.copy()
    res = orc.OI(Xa.copy(), Yw, Sa, So, regularization_on=True, forced_index=0)
    _check_pack(g, "onNone", res[:4], Yw)
    # the reference passes only direction='increasing' to KneeLocator (optimal_interpolation.py:37-38)
    assert list(g["kneed_kwargs"]) == ["direction=increasing"]


def test_error_averager(golden):
    g = golden("error_averager.npz")
    out = orc.error_averager(g["inp"])
    np.testing.assert_allclose(out, g["out"], rtol=1e-13, equal_nan=True)
    assert np.isnan(out[0, 0]) and np.isnan(out[2, 2])
    assert out[3, 3] == 0.5


class _Reader:
    pass


@pytest.mark.parametrize("tag", ["72x144_k5", "36x72_k9"])
def test_averaging(golden, tag):
    g = golden(f"averaging_{tag}.npz")
    r = _Reader()
    r.sat_data = syn.granule_stack(int(g["ny"]), int(g["nx"]), int(g["k"]), int(g["seed"]))
    res = orc.averaging("2019-06-01", "2019-07-01", r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
    for a, nm in zip(res[:5], ("sat_vcd", "sat_err", "ctm_vcd", "aux1", "aux2")):
        np.testing.assert_allclose(a, g[nm], rtol=1e-13, equal_nan=True)
    assert abs(res[5].timestamp() - float(g["avg_ts"])) < 1e-3


def test_upscaler(golden):
    g = golden("upscaler.npz")
    X, Y, Z, gs = g["X"], g["Y"], g["Z"], float(g["grid_size"])
    for tag in ("1x1", "10x10", "8x10", "pass"):
        for err in (False, True):
            k = f"{tag}_{'var' if err else 'mean'}"
            ctm = {"Latitude": g[k + "_clat"], "Longitude": g[k + "_clon"]}
            dlat = abs(ctm["Latitude"][0, 0] - ctm["Latitude"][1, 0])
            dlon = abs(ctm["Longitude"][0, 0] - ctm["Longitude"][0, 1])
            ox, oy, oz, need = orc.upscaler(X, Y, Z.copy(), ctm, gs, np.sqrt(dlat ** 2 + dlon ** 2), error=err)
            assert bool(need) == bool(g[k + "_need"])
            # convolve2d's accumulation order differs from ours: few-ulp slack
            np.testing.assert_allclose(oz, g[k + "_Z"], rtol=1e-12, equal_nan=True)
    np.testing.assert_allclose(np.ones((3, 4)) / 12, g["box_3_4"])
    np.testing.assert_allclose(np.ones((3, 4)) / 144, g["box2_3_4"])


def test_interpolator(golden):
    g = golden("interpolator.npz")
    s = syn.swath_granule(5005)
    for f in ("vcd", "amf", "uncertainty", "quality_flag", "latitude_center", "longitude_center"):
        np.testing.assert_array_equal(getattr(s, f), g["in_" + f])
    for tag in ("fine", "coarse"):
        ctm = {"Latitude": g[f"{tag}_clat"], "Longitude": g[f"{tag}_clon"]}
        for it in (4, 2, 1):
            r = orc.interpolator(it, float(g[f"{tag}_gs"]), s, ctm, 0.75, record_type=cfg.satellite_amf)
            assert r is not None and bool(r.ctm_upscaled_needed) == bool(g[f"{tag}_t{it}_need"])
            for f in ("vcd", "amf", "uncertainty", "latitude_center", "longitude_center"):
                np.testing.assert_allclose(np.asarray(getattr(r, f)), g[f"{tag}_t{it}_{f}"], rtol=1e-12, equal_nan=True)
    ctm = syn.regional_ctm_grid(-80.0, -60.0, 100.0, 140.0, 2.0, 2.5)
    assert orc.interpolator(4, 0.25, s, ctm, 0.75, record_type=cfg.satellite_amf) is None
    assert bool(g["miss_is_none"])


TIE_GRIDS = ("gs100_1x125", "gs100_2x25", "gs025_05x0625", "global_1x125")
L3_CASES = (("MOPITT", 6201, "gs100_1x125"), ("MOPITT", 6201, "gs100_2x25"), ("GOSAT", 6202, "gs100_2x25"))


def tie_case(g, tag):
    """(X, Y, ctm dict, grid_size, threshold, {name: Z}) of one exact-tie grid pair of upscaler_ties.npz; the fine grid is
    rebuilt the way interpolator() builds it (interpolator.py:136-143) and checked against the stored one."""
    la0, la1, lo0, lo1, dlat, dlon, gs = (float(v) for v in g[f"{tag}_spec"])
    ctm = {"Latitude": g[f"{tag}_clat"], "Longitude": g[f"{tag}_clon"]}
    X, Y = np.meshgrid(np.arange(ctm["Longitude"].min(), ctm["Longitude"].max() + gs, gs),
                       np.arange(ctm["Latitude"].min(), ctm["Latitude"].max() + gs, gs))
    np.testing.assert_array_equal(X, g[f"{tag}_X"])
    np.testing.assert_array_equal(Y, g[f"{tag}_Y"])
    fields = {"index": np.arange(X.size, dtype=np.float64).reshape(X.shape), "rand": g[f"{tag}_Zrand"]}
    return X, Y, ctm, gs, float(np.sqrt(dlat ** 2 + dlon ** 2)), fields


def count_exact_ties(X, Y, ctm):
    """model centres with two or more equidistant nearest fine nodes (regular grids: per-axis argmin multiplicity)"""
    def axis_ties(nodes, centres):
        d = np.abs(nodes[None, :] - centres[:, None])
        return (d == d.min(axis=1, keepdims=True)).sum(axis=1) > 1
    tx = axis_ties(X[0], ctm["Longitude"][0])
    ty = axis_ties(Y[:, 0], ctm["Latitude"][:, 0])
    return int((tx[None, :] | ty[:, None]).sum())


def check_tie_fields(g, tag, run):
    """run(Z, error) -> model-grid field; compared with the reference's _upscaler outputs: NaN pattern bit-equal, 1e-12"""
    for nm in ("index", "rand"):
        for err in (False, True):
            want = g[f"{tag}_{nm}_{'var' if err else 'mean'}"]
            got = run(nm, err)
            assert got.shape == want.shape
            assert np.array_equal(np.isnan(got), np.isnan(want)), (tag, nm, err)
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=0, equal_nan=True, err_msg=f"{tag} {nm} error={err}")


@pytest.mark.parametrize("tag", TIE_GRIDS)
def test_upscaler_exact_ties(golden, tag):
    """Model centres exactly midway between fine nodes (grid_size 1.0 vs 1.25 / 2.5 deg, reader.py:1209,:1271; 0.25 vs
    0.625): the pick among equidistant nodes is cKDTree's, and the oracle builds the same tree (interpolator.py:78-91)."""
    g = golden("upscaler_ties.npz")
    X, Y, ctm, gs, thr, fields = tie_case(g, tag)
    assert count_exact_ties(X, Y, ctm) > 0.2 * ctm["Latitude"].size          # the fixture really is about ties
    check_tie_fields(g, tag, lambda nm, err: orc.upscaler(X, Y, fields[nm].copy(), ctm, gs, thr, error=err)[2])


def check_l3_record(g, sensor, grid, it, r, rtol):
    assert isinstance(r, cfg.satellite_opt)
    for name in g[f"l3_{sensor}_{grid}_t{it}_arrays"]:
        want = g[f"l3_{sensor}_{grid}_t{it}_{name}"]
        got = np.asarray(getattr(r, str(name)))
        if want.shape == (1,):
            assert got.shape == (1,)
            continue
        assert got.shape == want.shape, (sensor, grid, it, name)
        assert np.array_equal(np.isnan(got), np.isnan(want)), (sensor, grid, it, name)
        np.testing.assert_allclose(got, want, rtol=rtol, atol=0, equal_nan=True, err_msg=f"{sensor} {grid} type {it} {name}")


@pytest.mark.parametrize("sensor,seed,grid", L3_CASES)
def test_interpolator_lattice_l3_ties(golden, sensor, seed, grid):
    """interpolator() on a level-3 lattice record (MOPITT MOP03 style, reader.py:1150-1211: grid_size 1.0, flag 0.0):
    every fine node is equidistant from four lattice centres (type 4 gathers through that tie) and every other model
    centre from two fine nodes."""
    g = golden("upscaler_ties.npz")
    ctm = {"Latitude": g[f"{grid}_clat"], "Longitude": g[f"{grid}_clon"]}
    s = syn.lattice_l3_granule(seed, sensor=sensor)
    for it in (1, 4):
        r = orc.interpolator(it, float(g[f"{grid}_spec"][6]), s, ctm, 0.0, record_type=cfg.satellite_opt)
        check_l3_record(g, sensor, grid, it, r, 1e-12)


LEVEL_KINDS = {"amf": 6101, "MOPITT": 6102, "GOSAT": 6103}


def level_granule_from_golden(g, kind):
    """the seeded swath record, checked field by field against the inputs stored with the reference's outputs"""
    s = syn.swath_level_granule(LEVEL_KINDS[kind], kind=kind, nz=3)
    for f in dataclasses.fields(s):
        v = getattr(s, f.name)
        if isinstance(v, np.ndarray) and v.size > 1:
            np.testing.assert_array_equal(v, g[f"{kind}_in_{f.name}"])
    return s


def check_level_record(g, kind, tag, it, r, rtol):
    """every array / flag the reference's interpolator() returned for this record kind"""
    want_type = cfg.satellite_amf if kind == "amf" else cfg.satellite_opt
    assert isinstance(r, want_type)
    for name in g[f"{kind}_{tag}_t{it}_arrays"]:
        want = g[f"{kind}_{tag}_t{it}_{name}"]
        got = getattr(r, str(name))
        if want.dtype.kind in "bU":
            assert got == want.item(), (kind, tag, it, name)
        elif want.shape == (1,):                     # np.empty((1)) placeholders: shape only (:180, :250)
            assert np.shape(got) == (1,), (kind, tag, it, name)
        else:
            assert np.shape(got) == want.shape, (kind, tag, it, name, np.shape(got), want.shape)
            np.testing.assert_allclose(np.asarray(got), want, rtol=rtol, equal_nan=True, err_msg=f"{kind} {tag} type {it} {name}")
    for name in ("profile", "latitude_corner", "longitude_corner", "quality_flag", "ctm_vcd", "ctm_time_at_sat"):
        if hasattr(r, name):
            assert getattr(r, name) == [], name      # positional [] slots of the rebuild (:285-290)
    assert r.time == datetime.datetime(2019, 6, 15, 13, 45)


@pytest.mark.parametrize("kind", ["amf", "MOPITT", "GOSAT"])
def test_interpolator_level_cubes(golden, kind):
    """3-D loops of interpolator(): scattering weights / pressure (satellite_amf, interpolator.py:191-213) and the
    satellite_opt branch (MOPITT, GOSAT; :216-291), against the reference's own outputs."""
    g = golden("interpolator_levels.npz")
    s = level_granule_from_golden(g, kind)
    rt = cfg.satellite_amf if kind == "amf" else cfg.satellite_opt
    for tag in ("coarse", "fine"):
        ctm = {"Latitude": g[f"{tag}_clat"], "Longitude": g[f"{tag}_clon"]}
        for it in (4, 1):
            r = orc.interpolator(it, float(g[f"{tag}_gs"]), s, ctm, 0.75, record_type=rt)
            check_level_record(g, kind, tag, it, r, 1e-12)


def test_interpolator_type3_rbf(golden):
    """RBFInterpolator(neighbors=5) restatement against the reference's own type-3 outputs."""
    g = golden("interpolator_rbf.npz")
    s = syn.swath_granule(5005)
    for tag, (dlat, dlon) in {"fine": (0.25, 0.25), "coarse": (2.0, 2.5)}.items():
        ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, dlat, dlon)
        r = orc.interpolator(3, 0.25, s, ctm, 0.75, record_type=cfg.satellite_amf)
        assert r is not None and bool(r.ctm_upscaled_needed) == bool(g[f"{tag}_t3_need"])
        for f in ("vcd", "amf", "uncertainty", "latitude_center", "longitude_center"):
            want = g[f"{tag}_t3_{f}"]
            got = np.asarray(getattr(r, f))
            assert np.array_equal(np.isnan(got), np.isnan(want))
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.nanmax(np.abs(want)), equal_nan=True)
    pts = np.column_stack((s.longitude_center.ravel(), s.latitude_center.ravel()))
    for only in (False, True):       # the HIP backend skips masked targets: same answer
        o = orc.interpolosis_rbf(pts, g["single_Z"], g["single_X"], g["single_Y"], g["single_dists"], 0.25, only_unmasked=only)
        np.testing.assert_allclose(o, g["single_out"], rtol=0, atol=1e-12 * np.nanmax(np.abs(g["single_out"])), equal_nan=True)


def test_interpolator_type3_rbf_on_a_lattice_follows_scipys_tree(golden):
    """Points on a regular lattice: many targets are equidistant from several candidates for the fifth neighbour, and the
    reference's answer is whichever one ``RBFInterpolator``'s own ``KDTree(y)`` returns (interpolator.py:21-27)."""
    g = golden("interpolator_rbf_ties.npz")
    for tag in ("centres", "nodes", "mesh"):
        want = g[f"{tag}_out"]
        got = orc.interpolosis_rbf(g["points"], g["Z"], g[f"{tag}_X"], g[f"{tag}_Y"], g[f"{tag}_dists"], 0.25)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max(), equal_nan=True)


L3_RBF_CASES = [("MOPITT", 6201, "gs100_1x125"), ("GOSAT", 6202, "gs100_2x25")]


def check_l3_rbf_record(g, sensor, grid, r, tol):
    """every array of the record against the reference's, to ``tol`` of the field's range (a thin-plate system on lattice
    points is conditioned ~1e3: LAPACK's and another elimination order differ by ~1e-13)"""
    assert isinstance(r, cfg.satellite_opt)
    for name in g[f"l3_{sensor}_{grid}_t3_arrays"]:
        want = g[f"l3_{sensor}_{grid}_t3_{name}"]
        got = np.asarray(getattr(r, str(name)))
        if want.shape == (1,):
            assert got.shape == (1,)
            continue
        assert got.shape == want.shape, (sensor, grid, name)
        assert np.array_equal(np.isnan(got), np.isnan(want)), (sensor, grid, name)
        np.testing.assert_allclose(got, want, rtol=0, atol=tol * np.nanmax(np.abs(want)), equal_nan=True, err_msg=f"{sensor} {grid} {name}")


@pytest.mark.parametrize("sensor,seed,grid", L3_RBF_CASES)
def test_interpolator_type3_on_lattice_l3_records(golden, sensor, seed, grid):
    """interpolator(3, ...) on level-3 lattice records: every fine node is the centre of four lattice points and has
    several equidistant candidates for its fifth neighbour."""
    g = golden("interpolator_rbf_ties.npz")
    ctm = {"Latitude": g[f"{grid}_clat"], "Longitude": g[f"{grid}_clon"]}
    r = orc.interpolator(3, 1.0, syn.lattice_l3_granule(seed, sensor=sensor), ctm, 0.0, record_type=cfg.satellite_opt)
    check_l3_rbf_record(g, sensor, grid, r, 1e-11)


@pytest.mark.parametrize("tag", ["m_eccoh", "m_gmi64", "m_up", "g_eccoh", "g_gmi64", "g_up"])
def test_ak_conv(golden, tag):
    """ak_conv_mopitt / ak_conv_gosat restatement against the reference's own outputs."""
    from amf_cases import akconv_cases, check_akconv_against_golden
    sensor, ctm, sat = akconv_cases()[tag]()
    with np.errstate(all="ignore"):
        res = orc.ak_conv(ctm, sat, sensor)
    check_akconv_against_golden(golden("ak_conv.npz"), tag, res, 1e-13)


@pytest.mark.parametrize("tag", ["eccoh", "gmi64", "up"])
def test_pwv(golden, tag):
    from amf_cases import pwv_cases, check_pwv_against_golden
    ctm, sat = pwv_cases()[tag]()
    check_pwv_against_golden(golden("pwv.npz"), tag, orc.pwv_calculator(ctm, sat), 1e-13)


def test_records_match_reference(golden):
    g = golden("records.npz")
    for nm in ("satellite_amf", "satellite_opt", "satellite_ssmis", "ctm_model"):
        ours = [f.name for f in dataclasses.fields(getattr(cfg, nm))]
        assert ours == list(g[nm])


def test_kneedle_properties():
    """PARITY UNPINNED (kneed absent).  Properties of the published algorithm only."""
    x = np.arange(0.1, 10, 0.1)
    y = x / (x + 1.5)                                   # concave increasing, like mean AK vs scale
    knee, idx = orc.kneedle_knee(x, y)
    xn = (x - x.min()) / (x.max() - x.min())
    yn = (y - y.min()) / (y.max() - y.min())
    assert idx == int(np.argmax(yn - xn)) and knee == x[idx]
    assert orc.kneedle_knee(x, 2 * x + 1) == (None, None)          # straight line: no knee
    assert orc.kneedle_knee(x, np.full_like(x, np.nan)) == (None, None)


def test_dense_limit_reduces_to_diag():
    """L -> 0 with H = selection: dense OI == element-wise OI at observed cells, Xa elsewhere."""
    c = syn.diag_case(18, 36, 120, 77)
    Yc = c.Y.copy()
    xb, ak, inc, err, _, _ = orc.OI(c.Xa.copy(), Yc, c.Sa, c.So, regularization_on=False)
    cells = np.flatnonzero(~np.isnan(Yc.ravel()))
    r = orc.dense_oi(c.lat, c.lon, c.Xa, c.Sa, c.lat.ravel()[cells], c.lon.ravel()[cells], cells,
                     Yc.ravel()[cells], c.So.ravel()[cells], L_km=1e-3, want_error=True)
    np.testing.assert_allclose(r["xa"][cells], xb.ravel()[cells], rtol=1e-12)
    un = np.setdiff1d(np.arange(c.Xa.size), cells)
    np.testing.assert_array_equal(r["xa"][un], c.Xa.ravel()[un])
    np.testing.assert_allclose(r["ak_obs"], ak.ravel()[cells], rtol=1e-10)
    np.testing.assert_allclose(r["err"][cells], err.ravel()[cells], rtol=1e-10)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_amf_recal(golden, tag):
    from amf_cases import amf_cases, check_against_golden
    ctm, sat = amf_cases()[tag]()
    check_against_golden(golden("amf_recal.npz"), tag, orc.amf_recal(ctm, sat), rtol=1e-12)
