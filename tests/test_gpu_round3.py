"""Round-3 GPU parity tests (run with -m gpu on an MI355X).

  * exact nearest-neighbour ties in _upscaler / interpolator() (interpolator.py:78-91): the reference's MOPITT / GOSAT
    settings (grid_size 1.0 against 1.25 / 2.5 degree model longitudes, reader.py:1209,:1271) put model centres exactly
    midway between fine nodes; the pick has to be cKDTree's.  Fixture: tests/golden/upscaler_ties.npz, outputs of the
    reference's own functions.

Tolerances: float64 regridding vs the reference 1e-12, NaN patterns bit-equal.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oi_oracle as orc                       # the checker (tests only)
from oisatgmi import _hip, synthetic as syn, config as cfg
from oisatgmi import interpolator as interp
from test_oracle_golden import (TIE_GRIDS, L3_CASES, tie_case, count_exact_ties, check_tie_fields, check_l3_record)


@pytest.fixture(scope="module")
def ctx():
    c = _hip.context()
    assert "gfx950" in c.device_info()["name"]
    return c


# ------------------------------------------------------------------------------------------------
# exact nearest-neighbour ties
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", TIE_GRIDS)
def test_upscaler_exact_ties_match_reference(ctx, golden, tag):
    g = golden("upscaler_ties.npz")
    X, Y, ctm, gs, thr, fields = tie_case(g, tag)
    nties = count_exact_ties(X, Y, ctm)
    assert nties > 0.2 * ctm["Latitude"].size
    check_tie_fields(g, tag, lambda nm, err: interp._upscaler(X, Y, fields[nm].copy(), ctm, gs, thr, error=err)[2])
    # the device search found exactly the tied model cells, and each of them went through the reference's tree
    plan = interp._upscale_plan(X, Y, ctm, gs, thr)
    assert plan.ties_resolved == nties


def test_nn_query_reports_exactly_the_tied_targets(ctx):
    """oisat_nn_query_ties on a 1-degree lattice: targets on nodes (unique), on edge midpoints (2-way), on cell centres
    (4-way), off-lattice (unique), beyond the radius (dropped, never reported) and a duplicated point (tie at distance 0)."""
    lon, lat = np.meshgrid(np.arange(0.0, 8.0), np.arange(0.0, 6.0))
    px = np.concatenate([lon.ravel(), [3.0]])            # point 48 duplicates node (0, 3) = index 3
    py = np.concatenate([lat.ravel(), [0.0]])
    tx = np.array([2.0, 2.5, 2.5, 2.3, 40.0, 3.0, 6.0, 6.4])
    ty = np.array([2.0, 2.0, 2.5, 2.9, 2.0, 0.0, 4.5, 4.75])
    want_tied = [1, 2, 5, 6]
    P, T = px.size, tx.size
    pb = ctx.upload(np.concatenate([px, py]))
    tb = ctx.upload(np.concatenate([tx, ty]))
    idx, ties = ctx.alloc(T * 4), ctx.alloc(T * 4)
    n = C.c_int64(-1)
    ctx.check(ctx.lib.oisat_nn_query_ties(ctx.h, pb.at(0), pb.at(P * 8), P, tb.at(0), tb.at(T * 8), T, 1.5, idx.ptr, None,
                                          ties.ptr, C.byref(n)))
    ctx.sync()
    got = np.sort(ctx.download(ties.ptr, (n.value,), np.int32))
    assert got.tolist() == want_tied
    i = ctx.download(idx.ptr, (T,), np.int32)
    assert i[0] == 2 * 8 + 2 and i[3] == 3 * 8 + 2 and i[4] == -1 and i[7] == 5 * 8 + 6
    # untied answers equal the tree's; tied ones are patched to the tree's by NNIndex
    nn = interp.NNIndex(px, py)
    d, j = nn.query(np.column_stack((tx, ty)), 1.5)
    from scipy.spatial import cKDTree
    dd, jj = cKDTree(np.column_stack((px, py))).query(np.column_stack((tx, ty)))
    keep = dd <= 1.5
    assert np.array_equal(j[keep], jj[keep]) and np.all(j[~keep] == -1)
    np.testing.assert_array_equal(d[keep], dd[keep])
    assert nn.ties_resolved == len(want_tied)


@pytest.mark.parametrize("sensor,seed,grid", L3_CASES)
def test_interpolator_lattice_l3_ties_match_reference(ctx, golden, sensor, seed, grid):
    g = golden("upscaler_ties.npz")
    ctm = {"Latitude": g[f"{grid}_clat"], "Longitude": g[f"{grid}_clon"]}
    s = syn.lattice_l3_granule(seed, sensor=sensor)
    for it in (1, 4):
        r = interp.interpolator(it, float(g[f"{grid}_spec"][6]), s, ctm, 0.0)
        check_l3_record(g, sensor, grid, it, r, 1e-12)


def test_interpolator_type2_gathers_through_the_same_ties(ctx, golden):
    """NearestNDInterpolator (type 2) is a cKDTree underneath: on the lattice record its output equals type 4's."""
    g = golden("upscaler_ties.npz")
    grid = "gs100_1x125"
    ctm = {"Latitude": g[f"{grid}_clat"], "Longitude": g[f"{grid}_clon"]}
    s = syn.lattice_l3_granule(6201, sensor="MOPITT")
    r2 = interp.interpolator(2, 1.0, s, ctm, 0.0)
    r2o = orc.interpolator(2, 1.0, s, ctm, 0.0, record_type=cfg.satellite_opt)
    check_l3_record(g, "MOPITT", grid, 4, r2, 1e-12)
    np.testing.assert_allclose(r2.vcd, r2o.vcd, rtol=1e-12, equal_nan=True)


# ------------------------------------------------------------------------------------------------
# oisatgmi.oi() in the spatial modes on what average() really hands over (driver.py:53-63,:108-114)
# ------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, sat_data):
        self.sat_data = sat_data


def _month_of_granules(ny, nx, k, seed, months=(6,)):
    out = []
    for mth in months:
        out += syn.granule_stack(ny, nx, k, seed + mth, year=2019, month=mth, with_none=True, coverage=0.12)
    return out


def _facade(ny, nx, k, seed, months, mode, **attrs):
    from oisatgmi.driver import oisatgmi
    o = oisatgmi()
    o.reader_obj = _Reader(_month_of_granules(ny, nx, k, seed, months))
    o.oi_mode, o.corr_length_km, o.tile_deg = mode, 500.0, 45.0
    for k_, v in attrs.items():
        setattr(o, k_, v)
    return o


@pytest.mark.parametrize("mode", ["dense", "tiled"])
def test_oi_spatial_modes_after_average_one_month(ctx, mode):
    """reader records -> average() -> oi(): the one-month window run/job.py:77-82 passes gives (ny, nx) fields; the
    spatial modes analyse them on the grid of the first granule and fill the four attributes of driver.py:110-114."""
    ny, nx = 36, 72
    o = _facade(ny, nx, 4, 8100, (6,), mode)
    o.average("2019-06-01", "2019-07-01", gasname="NO2")
    assert o.ctm_averaged_vcd.shape == (ny, nx)
    o.oi("OMI", error_ctm=50.0)
    xa, y, se = o.ctm_averaged_vcd, o.sat_averaged_vcd, o.sat_averaged_error
    observed = np.isfinite(y) & ~np.isnan(se) & np.isfinite(xa)
    assert observed.sum() > 200 and o.oi_info["nobs"] == int((observed & np.isfinite(se)).sum())
    for a in (o.ctm_averaged_vcd_corrected, o.ak_OI, o.increment_OI, o.error_OI):
        assert a.shape == (ny, nx) and a.dtype == np.float64
        assert np.isnan(a[~observed]).all() and np.isfinite(a[observed]).all()
    if mode == "dense":                 # the global analysis against the float64 oracle, error fields included
        lat, lon = syn.global_grid(ny, nx)
        use = observed & np.isfinite(se)
        cell = np.flatnonzero(use.ravel())
        s = o.oi_info["scale"]
        ref = orc.dense_oi(lat, lon, np.where(np.isfinite(xa), xa, 0.0), np.where(np.isfinite(xa), (0.5 * xa) ** 2, 0.0),
                           lat.ravel()[cell], lon.ravel()[cell], cell, y.ravel()[cell], se.ravel()[cell] ** 2, 500.0, scale=s,
                           want_error=True)
        fs = np.nanmax(np.abs(xa))
        assert np.abs(o.ctm_averaged_vcd_corrected.ravel()[cell] - ref["xa"][cell]).max() <= 1e-5 * fs
        assert np.abs(o.increment_OI.ravel()[cell] - ref["inc"][cell]).max() <= 1e-5 * fs
        np.testing.assert_allclose(o.error_OI.ravel()[cell], ref["err"][cell], rtol=2e-3, atol=1e-4 * fs)
        np.testing.assert_allclose(o.ak_OI.ravel()[cell], ref["ak_obs"], rtol=0, atol=2e-4)


@pytest.mark.parametrize("mode", ["dense", "tiled"])
def test_oi_spatial_modes_after_average_two_months(ctx, mode):
    """A two-month window: averaging() returns (ny, nx, 2) (averaging.py:53-58,:110-114; only the last month is reduced,
    the quirk of :97) and oi() runs one spatial analysis per month slice instead of tripping over the extra axis."""
    ny, nx = 36, 72
    o = _facade(ny, nx, 3, 8200, (6, 7), mode, oi_unobserved="xa")
    o.average("2019-06-01", "2019-08-01", gasname="NO2")
    assert o.ctm_averaged_vcd.shape == (ny, nx, 2)
    o.oi("OMI", error_ctm=50.0)
    assert len(o.oi_info["slices"]) == 2 and o.oi_info["slices"][0]["nobs"] == 0 and o.oi_info["slices"][1]["nobs"] > 200
    for a in (o.ctm_averaged_vcd_corrected, o.ak_OI, o.increment_OI, o.error_OI):
        assert a.shape == (ny, nx, 2)
    # the month that was reduced equals the one-month run of the same granules
    o1 = _facade(ny, nx, 3, 8200, (6, 7), mode, oi_unobserved="xa")
    o1.average("2019-07-01", "2019-08-01", gasname="NO2")
    o1.oi("OMI", error_ctm=50.0)
    for name in ("ctm_averaged_vcd_corrected", "ak_OI", "increment_OI", "error_OI"):
        np.testing.assert_array_equal(getattr(o, name)[:, :, 1], getattr(o1, name))
    # a grid that does not match the fields is refused with a clear message
    o1.grid_lat, o1.grid_lon = syn.global_grid(18, 36)
    with pytest.raises(ValueError, match="grid is"):
        o1.oi("OMI", error_ctm=50.0)


def test_oi_posterior_error_is_optional(ctx, monkeypatch):
    """OISAT_OI_ERROR=0 / oi_want_error=False: the n m^2-flop posterior error and averaging kernel are skipped (NaN),
    analysis and increment are unchanged."""
    ny, nx = 36, 72
    a = _facade(ny, nx, 4, 8100, (6,), "dense")
    a.average("2019-06-01", "2019-07-01")
    a.oi("OMI")
    monkeypatch.setenv("OISAT_OI_ERROR", "0")
    b = _facade(ny, nx, 4, 8100, (6,), "dense")
    b.average("2019-06-01", "2019-07-01")
    b.oi("OMI")
    assert b.oi_info["want_error"] is False and a.oi_info["want_error"] is True
    np.testing.assert_array_equal(a.ctm_averaged_vcd_corrected, b.ctm_averaged_vcd_corrected)
    np.testing.assert_array_equal(a.increment_OI, b.increment_OI)
    assert np.isnan(b.error_OI).all() and np.isnan(b.ak_OI).all() and np.isfinite(a.error_OI).any()


# ------------------------------------------------------------------------------------------------
# the gain solve stops refining when the float64 residual meets its tolerance (device-side test)
# ------------------------------------------------------------------------------------------------
def test_refinement_stops_at_its_tolerance(ctx):
    """oisat_gain_solve: at most `refine` rounds of { r = d - S z; stop if |r| <= tol |d|; z += M^-1 r }.  tol = 0 runs every
    round (residual down to rounding), the default 1e-6 stops as soon as it is met -- the reported list then repeats the
    last computed residual -- and a tolerance the plain solve already meets applies no correction at all.  The fields of
    the three agree far inside the 1e-5 bar, and with the float64 oracle."""
    from oisatgmi import dense
    p = syn.point_obs_case(90, 180, 3000, 7301, swaths=True)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    L = 500.0
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=3000, dtype=np.float32)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, y, p.obs_var)
    out = {}
    for name, tol in (("all", 0.0), ("default", None), ("none", 0.5)):
        resid = plan.run(L, refine=3, check_pd=True, want_resid=True, tol=tol)
        xa, inc = plan.download()
        out[name] = (resid, xa.copy(), plan.download_z())
        plan.run(L, refine=3, tol=tol)                          # the asynchronous path takes the same decisions
        xa2, _ = plan.download()
        np.testing.assert_array_equal(xa, xa2)
    r_all, r_def, r_none = out["all"][0], out["default"][0], out["none"][0]
    assert len(r_all) == len(r_def) == len(r_none) == 4
    assert r_all[0] > 1e-8 and r_all[-1] < 1e-11 and all(b < a for a, b in zip(r_all[:2], r_all[1:3]))
    assert r_all[0] == r_def[0] == r_none[0]                     # same plain solve
    k = next(i for i, r in enumerate(r_def) if r <= dense.REFINE_TOL)
    assert 1 <= k <= 2 and r_def[:k + 1] == r_all[:k + 1] and all(r == r_def[k] for r in r_def[k:])
    assert all(r == r_none[0] for r in r_none)                  # 0.5 |d| is met at once: no correction applied
    ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, p.obs_lat, p.obs_lon, cell, y, p.obs_var, L)
    scale = np.abs(ref["xa"]).max()
    assert np.abs(out["default"][1].ravel() - ref["xa"]).max() <= 1e-5 * scale
    assert np.abs(out["default"][1] - out["all"][1]).max() <= 1e-6 * scale
    dz = np.abs(out["default"][2] - out["all"][2]).max() / np.abs(out["all"][2]).max()
    assert dz <= 1e-4, dz


# ------------------------------------------------------------------------------------------------
# lock-step batches of >= 8 systems factor with leaf PAIRS (pair_mid_kernel / pair_panel_kernel)
# ------------------------------------------------------------------------------------------------
def test_batched_factorization_with_leaf_pairs(ctx):
    """oisat_batch_potrf on 10 systems (odd and even block counts, 1 .. 17 blocks): the recursion's leaves are pairs of
    block columns -- one pass over the rows below for TRSM / rank-128 update / TRSM -- and every member's factor agrees
    with oisat_potrf's (single-block leaves, another association of the same products) to fp32 rounding, solves through
    oisat_potrs, and a non-positive-definite member is reported by index whether the bad pivot sits in the first or the
    second block of a pair."""
    from oisatgmi import dense
    lib = ctx.lib
    ctx.check(lib.oisat_set_task_graph(ctx.h, 0))            # the recursion (tests/test_gpu_dag.py has the task graph's twin of this test)
    sizes = [2100, 1500, 1290, 1000, 777, 640, 300, 257, 129, 100]
    mats, refs = [], []
    for k, m in enumerate(sizes):
        p = syn.point_obs_case(72, 144, m, 9100 + k)
        cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
        mp = -(-m // 128) * 128
        oxyz = ctx.upload(dense.unit_vectors(p.obs_lat, p.obs_lon))
        osig = ctx.upload(np.sqrt(p.Sa.ravel())[cell], dtype=np.float64)
        ovar = ctx.upload(p.obs_var, dtype=np.float64)
        S1, S2 = ctx.alloc(mp * mp * 4), ctx.alloc(mp * mp * 4)
        for S in (S1, S2):
            ctx.check(lib.oisat_cov_build(ctx.h, oxyz.ptr, osig.ptr, ovar.ptr, m, dense.decay_constant(500.0), S.ptr, mp))
        info = C.c_int(-1)
        ctx.check(lib.oisat_potrf(ctx.h, S1.ptr, m, mp, C.byref(info)))
        refs.append(ctx.download(S1.ptr, (mp, mp), np.float32))
        mats.append((S2, ctx.alloc(mp * 128 * 4), m, mp))
    n = len(mats)
    Sp = (C.c_void_p * n)(*[a[0].ptr for a in mats])
    Tp = (C.c_void_p * n)(*[a[1].ptr for a in mats])
    mm = (C.c_int64 * n)(*[a[2] for a in mats])
    ld = (C.c_int64 * n)(*[a[3] for a in mats])
    bid = C.c_int(-1)
    ctx.check(lib.oisat_batch_create(ctx.h, n, Sp, mm, ld, Tp, C.byref(bid)))
    info2 = (C.c_int * 2)(-1, -1)
    ctx.check(lib.oisat_batch_potrf(ctx.h, bid.value, info2))
    assert list(info2) == [0, -1]
    for (S2, T, m, mp), ref in zip(mats, refs):
        got = ctx.download(S2.ptr, (mp, mp), np.float32)
        assert np.isfinite(np.tril(got)).all()
        assert np.abs(np.tril(got) - np.tril(ref)).max() <= 4e-6 * np.abs(np.tril(ref)).max(), m
        ctx.check(lib.oisat_factor_adopt(ctx.h, S2.ptr, m, mp, T.ptr))
        rhs = np.random.default_rng(m).normal(size=m)
        zb = ctx.upload(rhs)
        ctx.check(lib.oisat_potrs(ctx.h, S2.ptr, m, mp, zb.ptr))
        z = ctx.download(zb.ptr, (m,), np.float64)
        Lh = np.tril(ref[:m, :m]).astype(np.float64)
        zr = np.linalg.solve(Lh @ Lh.T, rhs)
        assert np.linalg.norm(z - zr) <= 1e-3 * np.linalg.norm(zr)
    ctx.check(lib.oisat_batch_destroy(ctx.h, bid.value))
    # bad pivots inside pairs: column 201 is in the second block of the pair (0, 1), column 300 in the first of (2, 3)
    for col in (200, 299):
        members = [(4.0 * np.eye(512) + 0.5).astype(np.float32) for _ in range(8)]
        members[5][col, col] = -1.0
        bufs = [ctx.upload(a) for a in members]
        tinvs = [ctx.alloc(512 * 128 * 4) for _ in members]
        Sp = (C.c_void_p * 8)(*[b.ptr for b in bufs])
        Tp = (C.c_void_p * 8)(*[b.ptr for b in tinvs])
        mm = (C.c_int64 * 8)(*[512] * 8)
        ld = (C.c_int64 * 8)(*[512] * 8)
        ctx.check(lib.oisat_batch_create(ctx.h, 8, Sp, mm, ld, Tp, C.byref(bid)))
        with pytest.raises(_hip.OisatError, match=f"matrix 5 not positive definite at column {col + 1}"):
            ctx.check(lib.oisat_batch_potrf(ctx.h, bid.value, info2))
        ctx.solve_status(clear=True)
        ctx.check(lib.oisat_batch_destroy(ctx.h, bid.value))
    ctx.check(lib.oisat_set_task_graph(ctx.h, -1))


def test_oi_dense_mode_through_the_facade_at_config2_size(ctx):
    """oisatgmi.oi(sensor, error_ctm) in `dense` mode at 360x720 with 1e4 observed cells -- the four
    attributes of driver.py:110-114, posterior error and averaging kernel included, against oracle.dense_oi on a subsample
    of the grid (every observed cell + 3000 others): analysis and increment 1e-6 of the field scale, error 2e-3, AK 5e-4."""
    from oisatgmi.driver import oisatgmi
    c = syn.diag_case(360, 720, 10000, 2001)
    lat, lon = syn.global_grid(360, 720)
    o = oisatgmi()
    o.ctm_averaged_vcd, o.sat_averaged_vcd = c.Xa.copy(), c.Y.copy()
    o.sat_averaged_error = np.sqrt(c.So)
    o.grid_lat, o.grid_lon = lat, lon
    o.oi_mode, o.corr_length_km, o.oi_unobserved = "dense", 500.0, "xa"
    o.oi("OMI", error_ctm=50.0)
    s = o.oi_info["scale"]
    assert o.oi_info["mode"] == "dense" and o.oi_info["want_error"] and o.oi_info["nobs"] > 9000
    Y = np.where(c.Y < 0, 0.0, c.Y)
    obs = np.isfinite(Y) & np.isfinite(c.So)
    cell = np.flatnonzero(obs.ravel())
    rng = np.random.default_rng(11)
    others = rng.choice(np.flatnonzero(~obs.ravel()), 3000, replace=False)
    sub = np.concatenate([cell, others])                        # the oracle's "grid": the observed cells first
    ref = orc.dense_oi(lat.ravel()[sub], lon.ravel()[sub], c.Xa.ravel()[sub], (0.5 * c.Xa.ravel()[sub]) ** 2,
                       lat.ravel()[cell], lon.ravel()[cell], np.arange(cell.size), Y.ravel()[cell], c.So.ravel()[cell], 500.0,
                       scale=s, want_error=True)
    fs = np.abs(c.Xa).max()
    # 1e-6, not 1e-5: with the increment's exponentials in double the fields sit at 4e-8 of the scale here (L = 500 km, gridded
    # observations: the case where rounds 1-2's v_exp_f32 left them 1.3e-5 off)
    assert np.abs(o.ctm_averaged_vcd_corrected.ravel()[sub] - ref["xa"]).max() <= 1e-6 * fs
    assert np.abs(o.increment_OI.ravel()[sub] - ref["inc"]).max() <= 1e-6 * fs
    np.testing.assert_allclose(o.error_OI.ravel()[sub], ref["err"], rtol=2e-3, atol=1e-4 * fs)
    np.testing.assert_allclose(o.ak_OI.ravel()[cell], ref["ak_obs"], rtol=0, atol=5e-4)   # diag(K H) from the fp32 factor alone
    assert (o.ak_OI.ravel()[others] == 0).all()                  # unobserved cells: averaging kernel 0 (OISAT_UNOBSERVED=xa)
    for a in (o.ctm_averaged_vcd_corrected, o.ak_OI, o.increment_OI, o.error_OI):
        assert a.shape == (360, 720) and np.isfinite(a).all()
