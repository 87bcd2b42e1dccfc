"""Parity of the HIP path against (a) the golden vectors written by the reference's own functions
and (b) the float64 CPU oracle, through the drop-in Python surface and the C-ABI underneath it.
Needs a real MI355X: run with  -m gpu.

Tolerances (written where used):
  float64 kernels vs float64 reference/oracle ........ rtol 1e-12 (operation order is the same;
                                                        only reduction orders differ by a few ulp)
  float32 kernels vs float64 reference ............... rtol 1e-5  (BASELINE.json north_star)
  dense analysis (fp32 factor + float64 refinement) .. 1e-5 of the field scale
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oi_oracle as orc                       # the checker (tests only)
from oisatgmi import _hip, synthetic as syn, config as cfg, dense
from oisatgmi.optimal_interpolation import OI
import oisatgmi.optimal_interpolation as oi_mod
from oisatgmi.averaging import averaging, error_averager
from oisatgmi.interpolator import interpolator, _upscaler, _interpolosis, NNIndex
from oisatgmi.driver import oisatgmi

FORCED_IDX = (0, 7, 37, 98)
RT64 = 1e-12
RT32 = 1e-5


@pytest.fixture(scope="module")
def ctx():
    c = _hip.context()
    info = c.device_info()
    assert "gfx950" in info["name"], info
    return c


def _oi_inputs(g):
    if "Xa" in g.files:
        return g["Xa"].copy(), g["Y"].copy(), g["Sa"].copy(), g["So"].copy()
    c = syn.diag_case(int(g["ny"]), int(g["nx"]), int(g["nobs"]), int(g["seed"]))
    Xa, Y, Sa, So = c.Xa.copy(), c.Y.copy(), c.Sa.copy(), c.So.copy()
    (i0, j0), (i1, j1), (i2, j2) = g["special"]
    Sa[i0, j0] = 0.0
    So[i1, j1] = np.inf
    Xa[i2, j2] = np.nan
    Sa[i2, j2] = np.nan
    return Xa, Y, Sa, So


def _check_pack(g, prefix, res, rtol, atol=0.0):
    stride = int(g["stride"])
    for nm, a in zip(("Xb", "AK", "inc", "err"), res):
        np.testing.assert_allclose(a.ravel()[::stride], g[f"{prefix}_{nm}"], rtol=rtol, atol=atol, equal_nan=True,
                                   err_msg=f"{prefix}_{nm}")
        assert int(np.isnan(a).sum()) == int(g[f"{prefix}_{nm}_nnan"]), f"{prefix}_{nm} NaN pattern"


# ------------------------------------------------------------------------------------------------
# OI  (optimal_interpolation.py:6-52)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["72x144", "360x720", "o3_72x144"])
def test_oi_float64_matches_reference(ctx, golden, tag):
    g = golden(f"oi_{tag}.npz")
    Xa, Y, Sa, So = _oi_inputs(g)
    Yw = Y.copy()
    res = OI(Xa.copy(), Yw, Sa, So, regularization_on=False)
    assert res[0].dtype == np.float64 and res[0].shape == Xa.shape
    _check_pack(g, "off", res, RT64)
    assert (Yw[~np.isnan(Yw)] >= 0).all()                                  # clamp happened in place
    np.testing.assert_array_equal(Yw.ravel()[::int(g["stride"])], g["Y_clamped"])
    for fi in FORCED_IDX:
        res = OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True, reg_index=fi)
        _check_pack(g, f"on{fi}", res, RT64)
    # the regularisation curve is pinned by the reference (captured at the KneeLocator call)
    curve = oi_mod.last_regularization["curve"]
    np.testing.assert_allclose(curve, g["curve_y"], rtol=1e-12)
    # knee pick (kneed: parity unpinned) -- must equal the oracle's restatement on the reference curve
    res = OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True)
    want = orc.kneedle_knee(g["curve_x"], g["curve_y"])[1]
    assert oi_mod.last_regularization["index"] == (0 if want is None else want)
    ref = orc.OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True)
    for a, b in zip(res, ref[:4]):
        np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True)


def test_oi_curve_is_bitwise_reproducible(ctx, golden):
    g = golden("oi_72x144.npz")
    Xa, Y, Sa, So = _oi_inputs(g)
    curves = []
    for _ in range(3):
        OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True)
        curves.append(oi_mod.last_regularization["curve"].copy())
    assert np.array_equal(curves[0], curves[1]) and np.array_equal(curves[0], curves[2])


@pytest.mark.parametrize("tag", ["72x144", "o3_72x144"])
def test_oi_float32_within_1e5(ctx, golden, tag):
    g = golden(f"oi_{tag}.npz")
    Xa, Y, Sa, So = (a.astype(np.float32) for a in _oi_inputs(g))
    res = OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=False)
    assert res[0].dtype == np.float32
    # increments are differences of O(1) numbers: tolerance relative to the field scale
    scale = float(np.nanmax(np.abs(g["off_Xb"])))
    _check_pack(g, "off", res, RT32, atol=RT32 * scale)
    res = OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True, reg_index=37)
    _check_pack(g, "on37", res, RT32, atol=RT32 * scale)
    np.testing.assert_allclose(oi_mod.last_regularization["curve"], g["curve_y"], rtol=RT32)


def test_oi_full_size_properties(ctx):
    """720x1440 (BASELINE config 3 grid): size-independent properties + oracle on a subsample."""
    c = syn.diag_case(720, 1440, 100000, 3001)
    Y0 = c.Y.copy()
    xb, ak, inc, err = OI(c.Xa.copy(), Y0, c.Sa, c.So, regularization_on=True)
    idx = oi_mod.last_regularization["index"]
    curve = oi_mod.last_regularization["curve"]
    obs = ~np.isnan(c.Y)
    assert np.isnan(xb[~obs]).all() and np.isfinite(xb[obs]).all()       # unobserved -> NaN
    np.testing.assert_allclose(xb[obs], (c.Xa + inc)[obs], rtol=0, atol=0)
    assert ((ak[obs] >= 0) & (ak[obs] <= 1)).all() and (np.diff(curve) > 0).all()
    # exact homogeneity under power-of-two rescaling: OI(4Xa,4Y,16Sa,16So) = 4*OI(...)
    xb4, ak4, inc4, err4 = OI(4 * c.Xa, 4 * np.where(c.Y < 0, 0, c.Y), 16 * c.Sa, 16 * c.So, regularization_on=True,
                              reg_index=idx)
    np.testing.assert_array_equal(xb4[obs], 4 * xb[obs])
    np.testing.assert_array_equal(ak4[obs], ak[obs])
    np.testing.assert_array_equal(err4[obs], 4 * err[obs])
    # oracle on a subsample of cells at the same index (the analysis is element-wise)
    sel = np.random.default_rng(0).choice(c.Xa.size, 20000, replace=False)
    f = orc.scaling_factors(True)[idx]
    o = orc.oi_fields(c.Xa.ravel()[sel], Y0.ravel()[sel], c.Sa.ravel()[sel], c.So.ravel()[sel], f)
    for a, b in zip((xb, ak, inc, err), o):
        np.testing.assert_allclose(a.ravel()[sel], b, rtol=RT64, equal_nan=True)
    np.testing.assert_allclose(curve, orc.oi_curve(c.Sa, c.So, orc.scaling_factors(True)), rtol=1e-12)


def test_fused_device_knee_matches_host_path(ctx, golden):
    """oisat_oi_fused picks the knee on the device; it must choose what the host pick chooses and
    produce the same fields as the two-call path -- on the golden cases and on curves of every kind."""
    from oisatgmi.optimal_interpolation import DiagOI
    from oisatgmi._kneedle import knee_index
    for tag in ("72x144", "o3_72x144"):
        g = golden(f"oi_{tag}.npz")
        Xa, Y, Sa, So = _oi_inputs(g)
        for dt in (np.float64, np.float32):
            d = DiagOI(Xa.size, dtype=dt)
            d.load(Xa, Y, Sa, So)
            idx_host, curve_host = d.run(True)
            ref = d.download(Xa.shape)
            d.load(Xa, Y, Sa, So)
            d.run_fused(True)
            idx_dev, curve_dev = d.fused_result()
            assert idx_dev == idx_host
            np.testing.assert_array_equal(curve_dev, curve_host)
            for a, b in zip(d.download(Xa.shape), ref):
                np.testing.assert_array_equal(a, b)
            d.run_fused(True, reg_index=37)
            assert d.fused_result()[0] == 37
            d.run_fused(False)
            assert d.fused_result()[0] == 0
    # knee pick alone on synthetic prior/obs-error mixes that move the knee across the sweep
    rng = np.random.default_rng(123)
    seen = set()
    for t in range(40):
        n = 4096
        Sa = rng.uniform(0.01, 10.0, size=n) ** rng.uniform(0.5, 3.0)
        So = rng.uniform(0.01, 10.0, size=n) * 10.0 ** rng.uniform(-2, 2)
        So[rng.uniform(size=n) < 0.3] = np.nan
        Xa = np.ones(n)
        d = DiagOI(n, dtype=np.float64)
        d.load(Xa, Xa, Sa, So)
        d.run_fused(True)
        idx_dev, curve = d.fused_result()
        k = knee_index(np.arange(0.1, 10, 0.1), curve)
        assert idx_dev == (0 if k is None else k)
        seen.add(idx_dev)
    assert len(seen) > 3


def test_fused_oi_is_hipgraph_capturable(ctx, golden):
    """include/oisat.h says oisat_oi_fused makes no host round trip: capture it in a hipGraph (through torch's stream
    capture on the stream the handle launches on), replay, and compare with the direct call."""
    import torch
    from oisatgmi.optimal_interpolation import DiagOI
    g = golden("oi_72x144.npz")
    Xa, Y, Sa, So = _oi_inputs(g)
    own = torch.cuda.Stream()
    ctx.set_stream(own.cuda_stream)
    try:
        d = DiagOI(Xa.size, dtype=np.float64, ctx=ctx)
        d.load(Xa, Y.copy(), Sa, So)
        d.run_fused(True)                                   # warm: uploads the scaling table once
        ctx.sync()
        idx0, curve0 = d.fused_result()
        direct = [a.copy() for a in d.download(Xa.shape)]
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=own):
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            d.run_fused(True)
        ctx.set_stream(own.cuda_stream)
        d.load(Xa, Y.copy(), Sa, So)                        # fresh inputs in the same buffers, then replay only
        graph.replay()
        torch.cuda.synchronize()
        idx1, curve1 = d.fused_result()
        assert idx1 == idx0
        np.testing.assert_array_equal(curve1, curve0)
        for a, b in zip(d.download(Xa.shape), direct):
            np.testing.assert_array_equal(a, b)
    finally:
        ctx.set_stream(None)


def test_oi_edge_shapes(ctx):
    # 1 cell, odd sizes, all-NaN observations
    for shape in ((1, 1), (3, 5), (1, 129), (257, 3)):
        rng = np.random.default_rng(sum(shape))
        Xa = rng.uniform(1, 5, size=shape)
        Y = rng.uniform(-1, 5, size=shape)
        Sa = (0.5 * Xa) ** 2
        So = rng.uniform(0.01, 1, size=shape)
        ref = orc.OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True)
        res = OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True)
        assert oi_mod.last_regularization["index"] == ref[5]
        for a, b in zip(res, ref[:4]):
            np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True)
    Xa = np.ones((4, 6))
    nanf = np.full((4, 6), np.nan)
    res = OI(Xa.copy(), nanf.copy(), Xa.copy(), nanf.copy(), regularization_on=True)
    assert oi_mod.last_regularization["index"] == 0 and all(np.isnan(a).all() for a in res)


def test_oi_special_values_against_oracle(ctx):
    """NaN / +-inf / 0 / negative / tiny values sprinkled into all four inputs: every output cell and its NaN-ness must
    follow the reference arithmetic (optimal_interpolation.py:14-52: in-place clamp, 0/0 -> NaN AK with K = 0, inf
    errors, ...), for a forced regularisation index and with regularisation off."""
    specials = np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, -3.0, 1e-300, 1e300, 5e-324])
    for seed in range(6):
        rng = np.random.default_rng(900 + seed)
        shape = (int(rng.integers(1, 40)), int(rng.integers(1, 70)))
        arrs = []
        for lo, hi in ((0.2, 10.0), (-1.0, 10.0), (0.01, 25.0), (0.01, 1.0)):          # Xa, Y, Sa, So
            a = rng.uniform(lo, hi, size=shape)
            hit = rng.uniform(size=shape) < 0.15
            a[hit] = rng.choice(specials, size=int(hit.sum()))
            arrs.append(a)
        Xa, Y, Sa, So = arrs
        with np.errstate(all="ignore"):
            for kw in (dict(regularization_on=False), dict(regularization_on=True, reg_index=37)):
                ref = orc.OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=kw["regularization_on"],
                             forced_index=kw.get("reg_index"))
                Yc = Y.copy()
                res = OI(Xa.copy(), Yc, Sa, So, **kw)
                for nm, a, b in zip(("Xb", "AK", "inc", "err"), res, ref[:4]):
                    assert np.array_equal(np.isnan(a), np.isnan(b)), (seed, kw, nm)
                    np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True, err_msg=f"seed {seed} {kw} {nm}")
                assert not (Yc[~np.isnan(Yc)] < 0).any()                                # clamped in the caller's array
    # the stacked reductions of averaging.py with the same specials
    stack = np.random.default_rng(7).uniform(0.1, 2.0, size=(7, 9, 11))
    hit = np.random.default_rng(8).uniform(size=stack.shape) < 0.25
    stack[hit] = np.random.default_rng(9).choice(specials[:6], size=int(hit.sum()))
    with np.errstate(all="ignore"):
        np.testing.assert_allclose(error_averager(stack.copy()), orc.error_averager(stack.copy()), rtol=RT64, equal_nan=True)


# ------------------------------------------------------------------------------------------------
# averaging.py
# ------------------------------------------------------------------------------------------------
def test_error_averager_matches_reference(ctx, golden):
    g = golden("error_averager.npz")
    out = error_averager(g["inp"])
    np.testing.assert_allclose(out, g["out"], rtol=RT64, equal_nan=True)
    out32 = error_averager(g["inp"].astype(np.float32))
    assert out32.dtype == np.float32
    np.testing.assert_allclose(out32, g["out"], rtol=RT32, equal_nan=True)


class _Reader:
    pass


@pytest.mark.parametrize("tag", ["72x144_k5", "36x72_k9"])
def test_averaging_matches_reference(ctx, golden, tag):
    g = golden(f"averaging_{tag}.npz")
    r = _Reader()
    r.sat_data = syn.granule_stack(int(g["ny"]), int(g["nx"]), int(g["k"]), int(g["seed"]))
    res = averaging("2019-06-01", "2019-07-01", r)
    for a, nm in zip(res[:5], ("sat_vcd", "sat_err", "ctm_vcd", "aux1", "aux2")):
        rt = 0.0 if nm != "sat_err" else RT64              # sequential-k sums are bit-identical to np.nanmean
        np.testing.assert_allclose(a, g[nm], rtol=rt, atol=0, equal_nan=True, err_msg=nm)
    assert abs(res[5].timestamp() - float(g["avg_ts"])) < 1e-3


def test_averaging_large_stack_against_oracle(ctx):
    r = _Reader()
    r.sat_data = syn.granule_stack(180, 360, 31, 77)
    res = averaging("2019-06-01", "2019-07-01", r)
    ref = orc.averaging("2019-06-01", "2019-07-01", r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
    for a, b in zip(res[:5], ref[:5]):
        np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True)


def test_averaging_edge_cases(ctx):
    """ragged / degenerate stacks: one granule, odd cell counts (scalar kernel path), float32 granules,
    cells never observed, a month with a single valid record among Nones."""
    rng = np.random.default_rng(17)
    for (ny, nx, k) in ((1, 1, 1), (3, 5, 2), (7, 9, 1), (33, 31, 6)):
        r = _Reader()
        r.sat_data = syn.granule_stack(ny, nx, k, 100 + ny, with_none=(k > 1), coverage=0.5) if ny * nx > 8 else None
        if r.sat_data is None:
            lat, lon = np.zeros((ny, nx)), np.zeros((ny, nx))
            import datetime
            t = datetime.datetime(2019, 6, 3, 12)
            r.sat_data = [None, cfg.satellite_amf(np.full((ny, nx), 2.0), np.empty(1), t, np.empty(1), lat, lon, [], [],
                                                  np.full((ny, nx), 0.5), [], np.empty(1), np.empty(1), False,
                                                  np.full((ny, nx), 3.0), t, np.full((ny, nx), 1.0), np.full((ny, nx), 1.5)), None]
        res = averaging("2019-06-01", "2019-07-01", r)
        ref = orc.averaging("2019-06-01", "2019-07-01", r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
        for a, b in zip(res[:5], ref[:5]):
            np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True)
    # float32 granules are reduced in float32, like np.nanmean of a float32 stack
    r = _Reader()
    r.sat_data = syn.granule_stack(24, 40, 7, 555)
    for g in r.sat_data:
        if g is not None:
            for f in ("vcd", "uncertainty", "ctm_vcd", "new_amf", "old_amf"):
                setattr(g, f, getattr(g, f).astype(np.float32))
    res = averaging("2019-06-01", "2019-07-01", r)
    ref = orc.averaging("2019-06-01", "2019-07-01", r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
    for a, b in zip(res[:5], ref[:5]):
        np.testing.assert_allclose(a, b, rtol=RT32, equal_nan=True)
    # error_averager: k = 1, k = 450 (an OMI month of orbits), all-NaN, all-inf
    e = rng.uniform(0.01, 1, size=(450, 9, 11))
    e[rng.uniform(size=e.shape) < 0.7] = np.nan
    e[:, 0, 0] = np.nan
    e[:, 1, 1] = np.inf
    np.testing.assert_allclose(error_averager(e), orc.error_averager(e), rtol=RT64, equal_nan=True)
    np.testing.assert_allclose(error_averager(e[:1]), orc.error_averager(e[:1]), rtol=RT64, equal_nan=True)


def test_averaging_full_size_month(ctx):
    """720 x 1440 x 30 daily composites (BASELINE grid): against the oracle on the whole field."""
    r = _Reader()
    r.sat_data = syn.granule_stack(720, 1440, 30, 3030, with_none=True, coverage=0.25)
    res = averaging("2019-06-01", "2019-07-01", r)
    ref = orc.averaging("2019-06-01", "2019-07-01", r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt)
    for a, b, nm in zip(res[:5], ref[:5], ("sat_vcd", "sat_err", "ctm_vcd", "aux1", "aux2")):
        np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True, err_msg=nm)
    assert res[0].shape == (720, 1440) and np.isnan(res[0]).any() and np.isfinite(res[0]).any()


def test_regrid_edge_cases(ctx):
    """granules that the readers really produce: everything flagged bad, NaN geolocation, a single scan line."""
    ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, 2.0, 2.5)
    s = syn.swath_granule(31)
    s.quality_flag[:] = 0.1                                     # all values masked -> "doesn't fall into the region"
    assert interpolator(4, 0.25, s, ctm, 0.75) is None
    s = syn.swath_granule(32)
    s.latitude_center[5:9, :] = np.nan                          # pixels without geolocation can never be neighbours
    s.longitude_center[5:9, :] = np.nan
    r = interpolator(4, 0.25, s, ctm, 0.75)
    rows = np.r_[0:5, 9:s.vcd.shape[0]]                          # the same granule without those scan lines
    clean = cfg.satellite_amf(s.vcd[rows], s.amf[rows], s.time, np.empty(1), s.latitude_center[rows], s.longitude_center[rows],
                              [], [], s.uncertainty[rows], s.quality_flag[rows], np.empty(1), np.empty(1), False, [], [], [], [])
    o = orc.interpolator(4, 0.25, clean, ctm, 0.75, record_type=cfg.satellite_amf)
    for f in ("vcd", "amf", "uncertainty"):
        np.testing.assert_allclose(getattr(r, f), getattr(o, f), rtol=RT64, equal_nan=True)
    s = syn.swath_granule(33, nscan=1, npix=200, lat0=10.0, lat1=10.0)
    r = interpolator(2, 0.25, s, ctm, 0.75)
    o = orc.interpolator(2, 0.25, s, ctm, 0.75, record_type=cfg.satellite_amf)
    assert (r is None) == (o is None)
    if r is not None:
        np.testing.assert_allclose(r.vcd, o.vcd, rtol=RT64, equal_nan=True)


def test_driver_methods(ctx, golden):
    g = golden("averaging_72x144_k5.npz")
    o = oisatgmi()
    o.reader_obj = _Reader()
    o.reader_obj.sat_data = syn.granule_stack(int(g["ny"]), int(g["nx"]), int(g["k"]), int(g["seed"]))
    o.average("2019-06-01", "2019-07-01", gasname="NO2")
    np.testing.assert_array_equal(o.sat_averaged_vcd, g["sat_vcd"])
    np.testing.assert_array_equal(o.ctm_averaged_vcd, g["ctm_vcd"])
    o.bias_correct("OMI", "NO2")
    np.testing.assert_array_equal(o.sat_averaged_vcd, orc.bias_correct(g["sat_vcd"], "OMI", "NO2"))
    sat_before = o.sat_averaged_vcd.copy()
    o.oi("OMI", error_ctm=50.0)
    Xa, Y, Sa, So = orc.driver_oi_inputs(g["ctm_vcd"], sat_before.copy(), g["sat_err"], g["aux1"], g["aux2"], "OMI", 50.0)
    ref = orc.OI(Xa.copy(), Y, Sa, So, regularization_on=True)
    for a, b in zip((o.ctm_averaged_vcd_corrected, o.ak_OI, o.increment_OI, o.error_OI), ref[:4]):
        np.testing.assert_allclose(a, b, rtol=RT64, equal_nan=True)
    assert (o.sat_averaged_vcd[~np.isnan(o.sat_averaged_vcd)] >= 0).all()      # OI clamps the attribute in place
    # O3 unit conversion and the GOSAT argument swap
    o2 = oisatgmi()
    o2.reader_obj = o.reader_obj
    o2.average("2019-06-01", "2019-07-01", gasname="O3")
    np.testing.assert_array_equal(o2.ctm_averaged_vcd, g["ctm_vcd"] / (2.69e16 * 1e-15))
    o2.oi("GOSAT", error_ctm=30.0)
    Xa, Y, Sa, So = orc.driver_oi_inputs(None, None, g["sat_err"], g["aux1"].copy(), g["aux2"], "GOSAT", 30.0)
    ref = orc.OI(Xa.copy(), Y, Sa, So, regularization_on=True)
    np.testing.assert_allclose(o2.ctm_averaged_vcd_corrected, ref[0], rtol=RT64, equal_nan=True)
    with pytest.raises(NotImplementedError):
        o.reporting("x", "NO2")
    with pytest.raises(NotImplementedError):
        o.read_data("GMI", ".", ["NO2"], "3-hourly", "OMI_NO2", ".", "201906")
    # output stage (driver.py:156-227): scaling-factor rule NaN/inf/0 -> 1 and the variable list
    post, prior = o.ctm_averaged_vcd_corrected.copy(), o.ctm_averaged_vcd.copy()
    obs = np.argwhere(np.isfinite(post))
    o.ctm_averaged_vcd_corrected[tuple(obs[0])] = 0.0                     # exact 0 -> 1
    o.ctm_averaged_vcd[tuple(obs[1])] = 0.0                               # x/0 = inf -> 1
    with np.errstate(all="ignore"):
        want = o.ctm_averaged_vcd_corrected / o.ctm_averaged_vcd
    want[np.isnan(want) | np.isinf(want) | (want == 0.0)] = 1.0
    np.testing.assert_array_equal(o.scaling_factor(), want)
    import tempfile
    from scipy.io import netcdf_file
    with tempfile.TemporaryDirectory() as td:
        path = o.write_to_nc("NO2_201906", td)
        nc = netcdf_file(path, "r", mmap=False)
        assert set(nc.variables) == {"time", "sat_averaged_vcd", "ctm_averaged_vcd_prior", "ctm_averaged_vcd_posterior",
                                     "sat_averaged_error", "ak_OI", "error_OI", "scaling_factor", "lon", "lat", "aux1", "aux2"}
        np.testing.assert_array_equal(nc.variables["scaling_factor"][:], want.astype(np.float32))
        assert nc.variables["ak_OI"][:].dtype.itemsize == 4 and nc.variables["lat"].shape == post.shape
        assert b"".join(nc.variables["time"][:]).decode().startswith("2019-06")
        nc.close()


# ------------------------------------------------------------------------------------------------
# amf_recal.py
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_amf_recal_matches_reference(ctx, golden, tag):
    from amf_cases import amf_cases, check_against_golden
    from oisatgmi.amf_recal import amf_recal
    ctm, sat = amf_cases()[tag]()
    # float64 model cubes: the reference's operation order, only log() may differ by an ulp -> 1e-11.
    # float32 model cubes (cases a, c): NumPy evaluates np.log(ctm_p) in float32 with a few-ulp SIMD
    # routine; the device uses the correctly rounded float32 log, and an ulp of log p (4e-7) times the
    # slope of the scattering-weight profile is what is left: 1e-5.
    rtol = 1e-11 if tag in ("b", "d") else 1e-5
    check_against_golden(golden("amf_recal.npz"), tag, amf_recal(ctm, sat), rtol=rtol)


def test_amf_recal_larger_against_oracle(ctx):
    from oisatgmi.amf_recal import amf_recal
    import copy
    ctm = syn.ctm_days(40, 60, 72, 2, 777, averaged=False, dtype=np.float64)
    sat = syn.amf_granules(ctm, 35, 3, 778, with_sw=True, with_trop=True)
    sat[0].pressure_mid[3, 5, 7] = np.nan                       # a broken satellite level
    sat[0].scattering_weights[:, 2, 2] = 0.0
    ref = orc.amf_recal(ctm, copy.deepcopy(sat))
    got = amf_recal(ctm, sat)
    o = oisatgmi()
    for a, b in zip(got, ref):
        if a is None:
            assert b is None
            continue
        for f in ("vcd", "ctm_vcd", "new_amf"):
            np.testing.assert_allclose(getattr(a, f), getattr(b, f), rtol=1e-10, equal_nan=True, err_msg=f)


# ------------------------------------------------------------------------------------------------
# interpolator.py
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# averaging-kernel convolution (ak_conv_mopitt.py / ak_conv_gosat.py; driver.conv_ak)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["m_eccoh", "m_gmi64", "m_up", "g_eccoh", "g_gmi64", "g_up"])
def test_ak_conv_matches_reference(ctx, golden, tag):
    """Outputs of the reference's own ak_conv_mopitt / ak_conv_gosat (tests/golden/ak_conv.npz).  float64 model cubes:
    same operation order, 1e-12.  float32 model cubes: NumPy evaluates np.log / np.log10 of float32 with few-ulp SIMD
    routines, the device with the correctly rounded ones -> 1e-5 (as for amf_recal)."""
    from amf_cases import akconv_cases, check_akconv_against_golden
    from oisatgmi.ak_conv_mopitt import ak_conv_mopitt
    from oisatgmi.ak_conv_gosat import ak_conv_gosat
    sensor, ctm, sat = akconv_cases()[tag]()
    res = (ak_conv_mopitt if sensor == "MOPITT" else ak_conv_gosat)(ctm, sat)
    assert res[1] is None
    check_akconv_against_golden(golden("ak_conv.npz"), tag, res, 1e-12 if tag.endswith("gmi64") else 1e-5)


@pytest.mark.parametrize("tag", ["eccoh", "gmi64", "up"])
def test_pwv_matches_reference(ctx, golden, tag):
    """pwv_calculator (pwv_cal.py) against the reference's own outputs; float32 model cubes are summed in float32 in
    the same order, so both dtypes agree to rounding."""
    from amf_cases import pwv_cases, check_pwv_against_golden
    from oisatgmi.pwv_cal import pwv_calculator
    ctm, sat = pwv_cases()[tag]()
    res = pwv_calculator(ctm, sat)
    assert res[1] is None
    check_pwv_against_golden(golden("pwv.npz"), tag, res, 1e-12 if tag != "eccoh" else 1e-6)
    # SSMIS branch of run/job.py: cal_pwv -> average -> oi; records of the third kind get NaN aux fields (averaging.py:89-91)
    o = oisatgmi()
    o.reader_obj = type("RO", (), {})()
    o.reader_obj.ctm_data, o.reader_obj.sat_data = pwv_cases()[tag]()
    o.cal_pwv()
    o.average("2019-05-01", "2019-06-01")
    assert np.isnan(o.aux1).all() and np.isfinite(o.ctm_averaged_vcd).any()
    o.oi("SSMIS", error_ctm=30.0)
    assert np.isfinite(o.ctm_averaged_vcd_corrected).any()


def test_ak_conv_then_average_and_oi(ctx):
    """The optimal-estimation branch of run/job.py end to end on the device: conv_ak -> average -> oi('GOSAT'),
    against the oracle's restatement of the same chain."""
    import copy
    from amf_cases import akconv_cases
    sensor, ctm, sat = akconv_cases()["g_eccoh"]()
    for s in sat:
        if s is not None:
            s.time = s.time.replace(month=5)
    ref_sat = orc.ak_conv(copy.deepcopy(ctm), copy.deepcopy(sat), "GOSAT")
    o = oisatgmi()

    class R:
        pass
    o.reader_obj = R()
    o.reader_obj.ctm_data, o.reader_obj.sat_data = ctm, sat
    o.conv_ak("GOSAT")
    o.average("2019-05-01", "2019-06-01")
    want = orc.averaging("2019-05-01", "2019-06-01", type("RO", (), {"sat_data": ref_sat})(), cfg.satellite_amf, cfg.satellite_opt)
    np.testing.assert_allclose(o.aux2, want[4], rtol=1e-5, equal_nan=True)             # mean model XCH4 (ctm_xcol)
    np.testing.assert_allclose(o.aux1, want[3], rtol=1e-12, equal_nan=True)            # mean observed XCH4 (x_col)
    o.oi("GOSAT", error_ctm=20.0)
    Xa, Y, Sa, So = orc.driver_oi_inputs(None, None, want[1], want[3].copy(), want[4], "GOSAT", 20.0)
    ref = orc.OI(Xa.copy(), Y, Sa, So, regularization_on=True)
    assert np.isfinite(o.ctm_averaged_vcd_corrected).any()
    np.testing.assert_allclose(o.ctm_averaged_vcd_corrected, ref[0], rtol=1e-5, equal_nan=True)


def test_upscaler_matches_reference(ctx, golden):
    g = golden("upscaler.npz")
    X, Y, Z, gs = g["X"], g["Y"], g["Z"], float(g["grid_size"])
    for tag in ("1x1", "10x10", "8x10", "pass"):
        for err in (False, True):
            k = f"{tag}_{'var' if err else 'mean'}"
            ctm = {"Latitude": g[k + "_clat"], "Longitude": g[k + "_clon"]}
            dlat = abs(ctm["Latitude"][0, 0] - ctm["Latitude"][1, 0])
            dlon = abs(ctm["Longitude"][0, 0] - ctm["Longitude"][0, 1])
            ox, oy, oz, need = _upscaler(X, Y, Z.copy(), ctm, gs, np.sqrt(dlat ** 2 + dlon ** 2), error=err)
            assert bool(need) == bool(g[k + "_need"])
            np.testing.assert_allclose(oz, g[k + "_Z"], rtol=RT64, equal_nan=True, err_msg=k)
            if not need:
                np.testing.assert_array_equal(ox, ctm["Longitude"])


def test_interpolator_matches_reference(ctx, golden):
    g = golden("interpolator.npz")
    s = syn.swath_granule(5005)
    for tag in ("fine", "coarse"):
        ctm = {"Latitude": g[f"{tag}_clat"], "Longitude": g[f"{tag}_clon"]}
        for it in (4, 2, 1):
            r = interpolator(it, float(g[f"{tag}_gs"]), s, ctm, 0.75)
            assert isinstance(r, cfg.satellite_amf)
            assert bool(r.ctm_upscaled_needed) == bool(g[f"{tag}_t{it}_need"])
            for f in ("vcd", "amf", "uncertainty", "latitude_center", "longitude_center"):
                np.testing.assert_allclose(np.asarray(getattr(r, f)), g[f"{tag}_t{it}_{f}"], rtol=RT64, equal_nan=True,
                                           err_msg=f"{tag} type {it} {f}")
    ctm = syn.regional_ctm_grid(-80.0, -60.0, 100.0, 140.0, 2.0, 2.5)
    assert interpolator(4, 0.25, s, ctm, 0.75) is None
    assert interpolator(1, 0.25, s, ctm, 0.75) is None
    assert interpolator(3, 0.25, s, ctm, 0.75) is None
    with pytest.raises(Exception, match="has not been implemented yet"):       # interpolator.py:34-36
        interpolator(5, 0.25, s, ctm, 0.75)
    # qhull cannot triangulate collinear pixels: the reference returns None for such a granule (:151-155)
    bad = syn.swath_granule(1, nscan=8, npix=1)
    bad.latitude_center = np.linspace(0, 7, 8)[:, None]
    bad.longitude_center = np.linspace(0, 7, 8)[:, None]
    assert interpolator(1, 0.25, bad, syn.regional_ctm_grid(-2.0, 10.0, -2.0, 10.0, 1.0, 1.0), 0.75) is None


# thin-plate-spline systems on 5 jittered pixels: the reference (LAPACK dgesv per neighbourhood) and the
# in-register LU agree to rounding x conditioning; measured 2.4e-14 of the field scale on the golden granule
RBF_TOL = 1e-11


def test_interpolator_type3_rbf_matches_reference(ctx, golden):
    """interpolator type 3 (interpolator.py:21-27, RBFInterpolator(neighbors=5)) against outputs of the
    reference's own function (tests/golden/interpolator_rbf.npz), NaN pattern included."""
    g = golden("interpolator_rbf.npz")
    s = syn.swath_granule(5005)
    for tag, (dlat, dlon) in {"fine": (0.25, 0.25), "coarse": (2.0, 2.5)}.items():
        ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, dlat, dlon)
        r = interpolator(3, 0.25, s, ctm, 0.75)
        assert isinstance(r, cfg.satellite_amf)
        assert bool(r.ctm_upscaled_needed) == bool(g[f"{tag}_t3_need"])
        for f in ("vcd", "amf", "uncertainty", "latitude_center", "longitude_center"):
            want = g[f"{tag}_t3_{f}"]
            got = np.asarray(getattr(r, f))
            assert np.array_equal(np.isnan(got), np.isnan(want)), f"{tag} {f}: NaN pattern"
            np.testing.assert_allclose(got, want, rtol=0, atol=RBF_TOL * np.nanmax(np.abs(want)), equal_nan=True,
                                       err_msg=f"{tag} type 3 {f}")
    # _interpolosis by itself: scattered targets, caller-supplied distances
    pts = np.column_stack((s.longitude_center.ravel(), s.latitude_center.ravel()))
    got = _interpolosis(pts, g["single_Z"], g["single_X"], g["single_Y"], 3, g["single_dists"], 0.25)
    want = g["single_out"]
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=0, atol=RBF_TOL * np.nanmax(np.abs(want)), equal_nan=True)


def test_rbf_edge_cases(ctx):
    """float32 kernels, fewer than 5 points, a singular (collinear) neighbourhood, a NaN value."""
    rng = np.random.default_rng(77)
    pts = rng.uniform(0.0, 4.0, size=(400, 2))
    Z = np.sin(pts[:, 0]) + pts[:, 1] ** 2
    X, Y = np.meshgrid(np.linspace(0.5, 3.5, 31), np.linspace(0.5, 3.5, 29))
    from scipy.spatial import cKDTree
    d, _ = cKDTree(pts).query(np.column_stack((X.ravel(), Y.ravel())))
    d = d.reshape(X.shape)
    want = orc.interpolosis_rbf(pts, Z, X, Y, d, 0.5)
    got = _interpolosis(pts, Z, X, Y, 3, d, 0.5)
    np.testing.assert_allclose(got, want, rtol=0, atol=RBF_TOL * np.abs(want).max())
    got32 = _interpolosis(pts, Z.astype(np.float32), X, Y, 3, d, 0.5)             # values follow the field dtype, solve is double
    assert got32.dtype == np.float32
    np.testing.assert_allclose(got32, want, rtol=0, atol=2e-6 * np.abs(want).max())
    Zn = Z.copy()
    Zn[5] = np.nan                                                               # poisons exactly the neighbourhoods that hold it
    wn = orc.interpolosis_rbf(pts, Zn, X, Y, d, 0.5)
    gn = _interpolosis(pts, Zn, X, Y, 3, d, 0.5)
    assert np.array_equal(np.isnan(gn), np.isnan(wn)) and np.isnan(wn).any()
    np.testing.assert_allclose(gn, wn, rtol=0, atol=RBF_TOL * np.abs(want).max(), equal_nan=True)
    # 4 points: scipy uses neighbors = min(5, P) = 4 -> a 7x7 system
    p4 = np.array([[0.0, 0.0], [1.0, 0.1], [0.2, 1.0], [1.1, 1.2]])
    z4 = np.array([1.0, 2.0, 3.0, 5.0])
    X4, Y4 = np.meshgrid(np.linspace(0.1, 1.0, 5), np.linspace(0.1, 1.0, 4))
    d4, _ = cKDTree(p4).query(np.column_stack((X4.ravel(), Y4.ravel())))
    d4 = d4.reshape(X4.shape)
    np.testing.assert_allclose(_interpolosis(p4, z4, X4, Y4, 3, d4, 5.0), orc.interpolosis_rbf(p4, z4, X4, Y4, d4, 5.0),
                               rtol=0, atol=1e-12)
    with pytest.raises(ValueError, match="At least 3 data points"):
        _interpolosis(p4[:2], z4[:2], X4, Y4, 3, d4, 5.0)
    # collinear neighbourhood: the monomial block is rank deficient -> LAPACK's zero pivot -> LinAlgError in scipy and here
    pl = np.column_stack((np.arange(6.0), np.zeros(6)))
    dl, _ = cKDTree(pl).query(np.column_stack((X4.ravel(), Y4.ravel())))
    with pytest.raises(np.linalg.LinAlgError):
        orc.interpolosis_rbf(pl, np.arange(6.0), X4, Y4, dl.reshape(X4.shape), 5.0)
    with pytest.raises(np.linalg.LinAlgError):
        _interpolosis(pl, np.arange(6.0), X4, Y4, 3, dl.reshape(X4.shape), 5.0)


def test_interpolosis_type1_standalone_matches_scipy(ctx):
    """_interpolosis(tri, Z, X, Y, 1, dists, thr) with a scipy Delaunay object, as the reference calls it."""
    from scipy.spatial import Delaunay, cKDTree
    from scipy.interpolate import LinearNDInterpolator
    rng = np.random.default_rng(21)
    pts = rng.uniform(0, 10, size=(3000, 2))
    Z = np.sin(pts[:, 0]) * np.cos(pts[:, 1])
    Z[rng.uniform(size=Z.size) < 0.02] = np.nan
    gx, gy = np.meshgrid(np.arange(-1, 11, 0.1), np.arange(-1, 11, 0.1))
    dists, _ = cKDTree(pts).query(np.stack([gx, gy], axis=-1))
    tri = Delaunay(pts)
    want = LinearNDInterpolator(tri, Z, fill_value=np.nan)((gx, gy))
    want[dists > 0.3 * 2.0] = np.nan
    got = _interpolosis(tri, Z, gx, gy, 1, dists, 0.3)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14, equal_nan=True)


def test_interpolator_with_levels_against_oracle(ctx):
    """scattering-weight / pressure levels (interpolator.py:191-209) are the same primitive batched."""
    s = syn.swath_granule(6006, nscan=120, npix=40)
    nz = 5
    rng = np.random.default_rng(1)
    s.scattering_weights = rng.uniform(0.1, 2.0, size=(nz,) + s.vcd.shape)
    s.pressure_mid = rng.uniform(100, 1000, size=(nz,) + s.vcd.shape)
    ctm = syn.regional_ctm_grid(-30.0, 50.0, -25.0, 45.0, 1.0, 1.25)
    r = interpolator(4, 0.25, s, ctm, 0.75)
    assert r.scattering_weights.shape == (nz,) + ctm["Latitude"].shape
    for z in (0, nz - 1):
        lev = cfg.satellite_amf(s.scattering_weights[z], s.pressure_mid[z], s.time, np.empty((1)), s.latitude_center,
                                s.longitude_center, [], [], s.uncertainty, s.quality_flag, np.empty((1)), np.empty((1)),
                                False, [], [], [], [])
        o = orc.interpolator(4, 0.25, lev, ctm, 0.75, record_type=cfg.satellite_amf)
        np.testing.assert_allclose(r.scattering_weights[z], o.vcd, rtol=RT64, equal_nan=True)
        np.testing.assert_allclose(r.pressure_mid[z], o.amf, rtol=RT64, equal_nan=True)


def test_nn_query_is_exact_within_radius(ctx):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(5)
    for P, T, R in ((1, 50, 0.5), (777, 3000, 0.3), (40000, 20000, 0.05), (5000, 5000, 2.0)):
        pts = rng.uniform(-10, 10, size=(P, 2))
        pts[::17] = np.nan if P > 20 else pts[::17]                   # NaN points are never neighbours
        tg = rng.uniform(-12, 12, size=(T, 2))
        d, i = NNIndex(pts[:, 0], pts[:, 1]).query(tg, max_dist=R)
        ok = ~np.isnan(pts).any(axis=1)
        tree = cKDTree(pts[ok])
        dr, ir = tree.query(tg)
        ir = np.flatnonzero(ok)[ir]
        inside = dr <= R
        np.testing.assert_array_equal(i[inside], ir[inside])
        np.testing.assert_allclose(d[inside], dr[inside], rtol=1e-15)
        assert (i[~inside] == -1).all() and np.isinf(d[~inside]).all()


def test_boxfilter_symm_against_oracle(ctx):
    rng = np.random.default_rng(8)
    lib = ctx.lib
    for (Ny, Nx, ky, kx) in ((33, 47, 3, 5), (20, 20, 10, 10), (7, 9, 16, 20), (64, 64, 1, 1), (5, 300, 2, 7)):
        Z = rng.normal(size=(Ny, Nx))
        Z[rng.uniform(size=Z.shape) < 0.02] = np.nan
        for var in (0, 1):
            zb = ctx.upload(Z)
            ob = ctx.alloc(Z.nbytes)
            ctx.check(lib.oisat_boxfilter_symm(ctx.h, _hip.F64, zb.ptr, Ny, Nx, ky, kx, var, ob.ptr))
            out = ctx.download(ob.ptr, Z.shape, np.float64)
            np.testing.assert_allclose(out, orc.boxfilter_symm(Z, ky, kx, bool(var)), rtol=1e-11, atol=1e-14, equal_nan=True)


# ------------------------------------------------------------------------------------------------
# dense Gaussian-B analysis (parity unpinned by the reference; oracle = float64 restatement)
# ------------------------------------------------------------------------------------------------
def _dense_case(ny, nx, m, seed, swaths=False):
    p = syn.point_obs_case(ny, nx, m, seed, swaths=swaths)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    return p, cell


@pytest.mark.parametrize("ny,nx,m,L", [(36, 72, 300, 800.0), (72, 144, 1000, 500.0), (90, 180, 2500, 300.0)])
def test_dense_analysis_against_oracle(ctx, ny, nx, m, L):
    p, cell = _dense_case(ny, nx, m, 1000 + m)
    ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y),
                       p.obs_var, L)
    for dt, tol in ((np.float64, 2e-6), (np.float32, 1e-5)):
        xb, inc, info = dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, L, refine=2, dtype=dt, tol=0.0,
                                       obs=dict(lat=p.obs_lat, lon=p.obs_lon, y=p.obs_y, var=p.obs_var))
        scale = np.abs(ref["xa"]).max()
        assert np.abs(xb.ravel() - ref["xa"]).max() <= tol * scale, (dt, np.abs(xb.ravel() - ref["xa"]).max() / scale)
        assert np.abs(inc.ravel() - ref["inc"]).max() <= tol * scale
        # gain solve: refinement drives the float64 residual of (HBH^T+R) z = d down
        assert info["residuals"][-1] < 1e-9 and info["residuals"][-1] < info["residuals"][0]
        # float64 fields: z is the float64 solution.  float32 fields: the innovation itself carries
        # the fp32 rounding of the background (6e-8 |Xa|), amplified by cond(S) ~ 1e2..1e3 in z
        ztol = 1e-9 if dt == np.float64 else 2e-5
        assert np.abs(info["z"] - ref["z"]).max() <= ztol * np.abs(ref["z"]).max()


def test_dense_analysis_is_bitwise_reproducible(ctx):
    """Same inputs, same bits: no atomics on the data path, fixed reduction orders, the pipelined triangular solves
    hand values over but never reorder sums.  Two runs of one plan and a run of a second plan must agree exactly."""
    p, cell = _dense_case(120, 240, 3000, 77, swaths=True)
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    outs = []
    for rep in range(2):
        plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=int(y.size), dtype=np.float32)
        plan.load_background(p.Xa, p.Sa)
        plan.load_obs(p.obs_lat, p.obs_lon, cell, y, p.obs_var)
        for _ in range(2):
            plan.run(350.0, refine=2)
            xa, inc = plan.download()
            outs.append((xa.copy(), inc.copy(), plan.download_z().copy()))
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            np.testing.assert_array_equal(a, b)


def test_dense_edge_sizes(ctx):
    """Observation counts around the 128-row blocking of the factorization, down to one and to none."""
    for m in (0, 1, 2, 5, 127, 128, 129, 257):
        p = syn.point_obs_case(36, 72, max(m, 1), 77 + m)
        obs = dict(lat=p.obs_lat[:m], lon=p.obs_lon[:m], y=p.obs_y[:m], var=p.obs_var[:m])
        xb, inc, info = dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, 600.0, refine=2, dtype=np.float32, obs=obs,
                                       want_error=True, tol=0.0)          # tol 0: every refinement round is run
        assert info["nobs"] == m and xb.dtype == np.float32 and xb.shape == p.Xa.shape
        if m == 0:                     # nothing observed: the analysis is the background
            np.testing.assert_array_equal(xb, p.Xa.astype(np.float32))
            assert not inc.any() and np.isnan(info["ak"]).all()
            np.testing.assert_allclose(info["err"], np.sqrt(p.Sa))
            continue
        cell = dense.regular_grid_cell(p.lat, p.lon, obs["lat"], obs["lon"])
        ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, obs["lat"], obs["lon"], cell, np.where(obs["y"] < 0, 0, obs["y"]),
                           obs["var"], 600.0)
        assert np.abs(xb.ravel() - ref["xa"]).max() <= 1e-6 * np.abs(ref["xa"]).max(), m
        assert info["residuals"][-1] < 1e-12


@pytest.mark.parametrize("L", [150.0, 400.0, 3000.0])
def test_latitude_window_equals_the_full_sum(ctx, L):
    """oisat_apply_increment / oisat_cov_residual with the latitude window (observations sorted by latitude) against the
    same entry points evaluating every pair: the skipped pairs are below 2^-64 of the largest term.  Observations
    include both poles and the date line; L = 3000 km makes the window wider than the sphere (no culling at all)."""
    lib = ctx.lib
    rng = np.random.default_rng(31)
    lat2, lon2 = syn.global_grid(90, 180)
    n = lat2.size
    m = 3000
    olat = np.concatenate([rng.uniform(-90, 90, m - 40), np.full(10, 89.9), np.full(10, -89.9), rng.uniform(-5, 5, 20)])
    olon = np.concatenate([rng.uniform(-180, 180, m - 20), np.tile([179.99, -179.99], 10)])
    order = np.argsort(olat, kind="stable")
    olat, olon = olat[order], olon[order]
    g = dense.decay_constant(L)
    gxyz = ctx.upload(dense.unit_vectors(lat2, lon2))
    gsig = ctx.upload(rng.uniform(0.5, 2.0, n))
    glat = ctx.upload(lat2.ravel())
    oxyz = ctx.upload(dense.unit_vectors(olat, olon))
    osig = ctx.upload(rng.uniform(0.5, 2.0, m))
    ovar = ctx.upload(rng.uniform(0.1, 1.0, m))
    z = ctx.upload(rng.normal(size=m))
    d = ctx.upload(rng.normal(size=m))
    olat_b = ctx.upload(olat)
    xb = ctx.upload(rng.uniform(1, 5, n))
    outs = []
    for win in (False, True):
        inc = ctx.alloc(n * 8)
        r = ctx.alloc(m * 8)
        ctx.check(lib.oisat_apply_increment(ctx.h, _hip.F64, gxyz.ptr, gsig.ptr, n, oxyz.ptr, osig.ptr, z.ptr, m, g, xb.ptr, None,
                                            inc.ptr, glat.ptr if win else None, olat_b.ptr if win else None))
        ctx.check(lib.oisat_cov_residual(ctx.h, oxyz.ptr, osig.ptr, ovar.ptr, m, g, d.ptr, z.ptr, r.ptr, olat_b.ptr if win else None))
        outs.append((ctx.download(inc.ptr, (n,), np.float64), ctx.download(r.ptr, (m,), np.float64)))
    (inc0, r0), (inc1, r1) = outs
    assert np.abs(inc0).max() > 0
    np.testing.assert_allclose(inc1, inc0, rtol=0, atol=1e-13 * np.abs(inc0).max())
    np.testing.assert_allclose(r1, r0, rtol=0, atol=1e-13 * np.abs(r0).max())
    # and against a float64 NumPy contraction of the same pairs
    C = orc.gaussian_corr(orc.unit_vectors(lat2.ravel()[::37], lon2.ravel()[::37]), orc.unit_vectors(olat, olon), L)
    want = ctx.download(gsig.ptr, (n,), np.float64)[::37] * (C @ (ctx.download(osig.ptr, (m,), np.float64) * ctx.download(z.ptr, (m,), np.float64)))
    np.testing.assert_allclose(inc1[::37], want, rtol=0, atol=3e-6 * np.abs(want).max())       # v_exp_f32 inside


@pytest.mark.parametrize("species", ["NO2", "HCHO", "O3"])
def test_dense_analysis_per_species(ctx, species):
    """BASELINE config 5 shapes (control_omino2 / control_omihcho / control_omio3.yml): the three parameter sets
    differ in ctm_error (50/50/10 %), value range (0-10, 0-20 x1e15 molec/cm2, 200-500 DU) and observation error
    (absolute vs 4 % of the column, reader.py:1035), i.e. in the conditioning of H B H^T + R."""
    L = 400.0
    p = syn.point_obs_case(90, 180, 2000, 5000 + len(species), swaths=True, species=species)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var, L)
    xb, inc, info = dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, L, refine=2, dtype=np.float32,
                                   obs=dict(lat=p.obs_lat, lon=p.obs_lon, y=p.obs_y, var=p.obs_var))
    scale = np.abs(ref["xa"]).max()
    assert np.abs(xb.ravel() - ref["xa"]).max() <= 1e-5 * scale, np.abs(xb.ravel() - ref["xa"]).max() / scale
    assert np.abs(inc.ravel() - ref["inc"]).max() <= 1e-5 * scale
    assert info["residuals"][-1] <= dense.REFINE_TOL          # the solve stops refining at this relative residual
    # the analysis moves towards the observations: smaller misfit at the observed cells than the background
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    assert np.abs(xb.ravel()[cell] - y).mean() < np.abs(p.Xa.ravel()[cell] - y).mean()


def test_dense_pieces_through_the_c_abi(ctx):
    """cov_build / potrf / potrs individually, ragged size (m not a multiple of the 128 block)."""
    lib = ctx.lib
    p, cell = _dense_case(72, 144, 777, 42)
    m, L = 777, 600.0
    mp = -(-m // 128) * 128
    g = dense.decay_constant(L)
    sb = np.sqrt(p.Sa.ravel())
    po = orc.unit_vectors(p.obs_lat, p.obs_lon)
    S_ref = orc.gaussian_corr(po, po, L) * sb[cell][:, None] * sb[cell][None, :]
    S_ref[np.diag_indices(m)] += p.obs_var
    oxyz = ctx.upload(dense.unit_vectors(p.obs_lat, p.obs_lon))
    osig = ctx.upload(sb[cell], dtype=np.float64)
    ovar = ctx.upload(p.obs_var, dtype=np.float64)
    S = ctx.alloc(mp * mp * 4)
    ctx.check(lib.oisat_cov_build(ctx.h, oxyz.ptr, osig.ptr, ovar.ptr, m, g, S.ptr, mp))
    Sh = ctx.download(S.ptr, (mp, mp), np.float32)
    tri = np.tril_indices(m)
    np.testing.assert_allclose(Sh[:m, :m][tri], S_ref[tri], rtol=2e-6, atol=1e-7 * S_ref.max())
    np.testing.assert_array_equal(Sh[m:, m:][np.tril_indices(mp - m)], np.eye(mp - m)[np.tril_indices(mp - m)])
    assert (Sh[m:, :m] == 0).all()
    info = C.c_int(-1)
    ctx.check(lib.oisat_potrf(ctx.h, S.ptr, m, mp, C.byref(info)))
    assert info.value == 0
    Lh = np.tril(ctx.download(S.ptr, (mp, mp), np.float32)[:m, :m]).astype(np.float64)
    rel = np.linalg.norm(Lh @ Lh.T - S_ref) / np.linalg.norm(S_ref)
    assert rel < 5e-7, rel
    rhs = np.random.default_rng(0).normal(size=m)
    zb = ctx.upload(rhs)
    ctx.check(lib.oisat_potrs(ctx.h, S.ptr, m, mp, zb.ptr))
    z = ctx.download(zb.ptr, (m,), np.float64)
    zr = np.linalg.solve(S_ref, rhs)
    assert np.linalg.norm(z - zr) / np.linalg.norm(zr) < 1e-3          # one fp32 solve, no refinement
    # a non-SPD matrix is reported, not silently factored
    bad = np.eye(256, dtype=np.float32)
    bad[200, 200] = -1.0
    Bb = ctx.upload(bad)
    with pytest.raises(_hip.OisatError):
        ctx.check(lib.oisat_potrf(ctx.h, Bb.ptr, 256, 256, C.byref(info)))
    assert info.value == 201


@pytest.mark.parametrize("schedule", ["recursive", "lookahead"])
def test_both_cholesky_schedules(ctx, schedule, monkeypatch):
    """The two schedules of oisat_potrf (recursive: large matrices; two-stream look-ahead: cache-sized ones)
    on the same ragged matrix: L L^T = S, same solution."""
    monkeypatch.setenv("OISAT_POTRF", schedule)
    lib = ctx.lib
    p, cell = _dense_case(72, 144, 1500, 4242)
    m, L = 1500, 500.0
    mp = -(-m // 128) * 128
    sb = np.sqrt(p.Sa.ravel())
    po = orc.unit_vectors(p.obs_lat, p.obs_lon)
    S_ref = orc.gaussian_corr(po, po, L) * sb[cell][:, None] * sb[cell][None, :]
    S_ref[np.diag_indices(m)] += p.obs_var
    oxyz = ctx.upload(dense.unit_vectors(p.obs_lat, p.obs_lon))
    osig = ctx.upload(sb[cell], dtype=np.float64)
    ovar = ctx.upload(p.obs_var, dtype=np.float64)
    S = ctx.alloc(mp * mp * 4)
    for rep in range(3):                                   # repeated calls reuse the events / aux stream
        ctx.check(lib.oisat_cov_build(ctx.h, oxyz.ptr, osig.ptr, ovar.ptr, m, dense.decay_constant(L), S.ptr, mp))
        info = C.c_int(-1)
        ctx.check(lib.oisat_potrf(ctx.h, S.ptr, m, mp, C.byref(info)))
        assert info.value == 0
        Lh = np.tril(ctx.download(S.ptr, (mp, mp), np.float32)[:m, :m]).astype(np.float64)
        rel = np.linalg.norm(Lh @ Lh.T - S_ref) / np.linalg.norm(S_ref)
        assert rel < 5e-7, (schedule, rep, rel)
    rhs = np.random.default_rng(1).normal(size=m)
    zb = ctx.upload(rhs)
    ctx.check(lib.oisat_potrs(ctx.h, S.ptr, m, mp, zb.ptr))
    z = ctx.download(zb.ptr, (m,), np.float64)
    zr = np.linalg.solve(S_ref, rhs)
    assert np.linalg.norm(z - zr) / np.linalg.norm(zr) < 2e-3


def test_dense_reduces_to_elementwise_oi_in_the_limit(ctx, golden):
    """L -> 0, H = cell selection: the dense path must reproduce the REFERENCE's OI at observed cells."""
    g = golden("oi_72x144.npz")
    Xa, Y, Sa, So = g["Xa"].copy(), g["Y"].copy(), g["Sa"].copy(), g["So"].copy()
    lat, lon = syn.global_grid(72, 144)
    ok = np.isfinite(Y) & np.isfinite(So) & np.isfinite(Xa) & np.isfinite(Sa)
    xb, inc, info = dense.OI_dense(Xa, Y.copy(), Sa, So, lat, lon, L_km=1e-3, refine=1, dtype=np.float64)
    want = g["off_Xb"].reshape(72, 144)
    # increments are formed in fp32 inside apply_increment: tolerance relative to the field scale
    assert np.abs(xb[ok] - want[ok]).max() <= 1e-6 * np.nanmax(np.abs(want))
    un = ~ok & np.isfinite(Xa)
    np.testing.assert_array_equal(xb[un], Xa[un])                       # no spread when L -> 0


def test_dense_posterior_error_and_gain_diag(ctx, golden):
    """The other two members of OI's 4-tuple for the dense analysis: sqrt(diag(B - B H^T S^-1 H B)) and
    diag(K H), both via MFMA TRSM of extra rows (oisat_posterior_error / oisat_gain_diag)."""
    for (ny, nx, m, L) in ((36, 72, 300, 800.0), (45, 90, 1100, 500.0)):
        p, cell = _dense_case(ny, nx, m, 900 + m)
        y = np.where(p.obs_y < 0, 0, p.obs_y)
        ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, p.obs_lat, p.obs_lon, cell, y, p.obs_var, L, want_error=True)
        xb, inc, info = dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, L, refine=2, dtype=np.float32, want_error=True,
                                       obs=dict(lat=p.obs_lat, lon=p.obs_lon, y=p.obs_y, var=p.obs_var))
        sig = np.sqrt(p.Sa).max()
        # err^2 = sig^2 - ||L^-1 HB||^2 is a difference of O(sig^2) numbers in fp32: absolute tolerance on err
        assert np.abs(info["err"].ravel() - ref["err"]).max() <= 2e-3 * sig, np.abs(info["err"].ravel() - ref["err"]).max() / sig
        np.testing.assert_allclose(info["ak_obs"], ref["ak_obs"], atol=2e-5, rtol=0)
        assert (info["err"] <= np.sqrt(p.Sa) * (1 + 1e-6)).all()            # analysis never less certain than the prior
    # reference-anchored limit: L -> 0, H = cell selection  =>  AK and sqrt(Sb) of optimal_interpolation.py:29-31,:52
    g = golden("oi_72x144.npz")
    Xa, Y, Sa, So = g["Xa"].copy(), g["Y"].copy(), g["Sa"].copy(), g["So"].copy()
    lat, lon = syn.global_grid(72, 144)
    ok = np.isfinite(Y) & np.isfinite(So) & np.isfinite(Xa) & np.isfinite(Sa)
    xb, inc, info = dense.OI_dense(Xa, Y.copy(), Sa, So, lat, lon, L_km=1e-3, refine=1, dtype=np.float64, want_error=True)
    ak_ref = g["off_AK"].reshape(72, 144)
    err_ref = g["off_err"].reshape(72, 144)
    pos = ok & (Sa > 0)                       # the reference's AK is NaN where Sa*reg == 0 (0/0)
    np.testing.assert_allclose(info["ak"][pos], ak_ref[pos], atol=2e-6, rtol=0)
    np.testing.assert_allclose(info["err"][pos], err_ref[pos], rtol=2e-3, atol=1e-4)
    un = ~ok & np.isfinite(Xa)
    np.testing.assert_allclose(info["err"][un], np.sqrt(Sa[un]), rtol=1e-6)       # unobserved: prior error untouched


def test_tiled_block_b_analysis(ctx):
    """Localised block-B (tiles + halo): each tile against the float64 oracle on the same observation set;
    with a halo that covers the globe every tile sees every observation and the result is the global one."""
    ny, nx, m, L = 36, 72, 900, 400.0
    p, cell = _dense_case(ny, nx, m, 4321)
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=60.0, halo_km=3 * L, dtype=np.float32)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    covered = np.zeros((ny, nx), dtype=int)
    for t in ta.tiles:
        covered[t["rows"][0]:t["rows"][1], t["cols"][0]:t["cols"][1]] += 1
    # two polar bands (their halo reaches the pole: one cap tile each, all longitudes) + 6 tiles in the middle band
    assert (covered == 1).all() and len(ta.tiles) == 1 + 6 + 1
    tu = dense.TiledAnalysis(p.lat, p.lon, tile_deg=60.0, halo_km=3 * L, dtype=np.float32, pool=ta.pool, merge_polar=False)
    tu.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    assert len(tu.tiles) == 3 * 6
    ta.run(L, refine=2, check_pd=True)
    xa, inc = ta.download()
    # merging a polar band into one cap tile changes nothing but the number of factorizations: same observation set,
    # same system, same increments as the band cut every 60 deg of longitude
    tu.run(L, refine=2, check_pd=True)
    xu, incu = tu.download()
    scale = np.abs(p.Xa).max()
    assert np.abs(inc - incu).max() <= 2e-6 * scale
    for t in ta.tiles:
        (y0, y1), (x0, x1) = t["rows"], t["cols"]
        o = t["obs"]
        # every observation inside the tile must be in its list; the halo adds more
        inside = (cell // nx >= y0) & (cell // nx < y1) & (cell % nx >= x0) & (cell % nx < x1)
        assert np.isin(np.flatnonzero(inside), o).all() and o.size >= inside.sum()
        sub = (slice(y0, y1), slice(x0, x1))
        # oracle on the tile: the innovation uses the global background at the obs cell
        lut = np.full(ny * nx, -1, dtype=np.int64)
        tile_cells = (np.arange(y0, y1)[:, None] * nx + np.arange(x0, x1)[None, :]).ravel()
        lut[tile_cells] = np.arange(tile_cells.size)
        sb = np.sqrt(p.Sa.ravel())
        po = orc.unit_vectors(p.obs_lat[o], p.obs_lon[o])
        S = orc.gaussian_corr(po, po, L) * sb[cell[o]][:, None] * sb[cell[o]][None, :]
        S[np.diag_indices_from(S)] += p.obs_var[o]
        z = np.linalg.solve(S, y[o] - p.Xa.ravel()[cell[o]])
        pg = orc.unit_vectors(p.lat[sub].ravel(), p.lon[sub].ravel())
        inc_ref = sb[tile_cells] * (orc.gaussian_corr(pg, po, L) @ (sb[cell[o]] * z))
        assert np.abs(inc[sub].ravel() - inc_ref).max() <= 1e-5 * scale
    # global halo == global analysis
    tb = dense.TiledAnalysis(p.lat, p.lon, tile_deg=90.0, halo_km=25000.0, dtype=np.float32)
    tb.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    assert all(t["obs"].size == m for t in tb.tiles)
    tb.run(L, refine=2)
    xg, _ = tb.download()
    ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, p.obs_lat, p.obs_lon, cell, y, p.obs_var, L)
    assert np.abs(xg.ravel() - ref["xa"]).max() <= 1e-5 * scale


def test_dense_config2_size_properties(ctx):
    """BASELINE config 2 (360x720, 1e4 obs): too big for a quick CPU solve of everything, so check
    the solve through its float64 residual, and the analysis against the oracle on a cell subsample."""
    p, cell = _dense_case(360, 720, 10000, 2002)
    L = 300.0
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=10000, dtype=np.float32)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, y, p.obs_var)
    resid = plan.run(L, refine=2, check_pd=True, want_resid=True)
    assert resid[-1] <= dense.REFINE_TOL, resid
    xa, inc = plan.download()
    z = plan.download_z()
    # oracle increment on 4000 random cells from OUR z would only test apply_increment; use the
    # oracle's own z from a CPU Cholesky of the same system (float64, ~10 s)
    sb = np.sqrt(p.Sa.ravel())
    po = orc.unit_vectors(p.obs_lat, p.obs_lon)
    S = orc.gaussian_corr(po, po, L) * sb[cell][:, None] * sb[cell][None, :]
    S[np.diag_indices_from(S)] += p.obs_var
    import scipy.linalg as sla
    zr = sla.cho_solve(sla.cho_factor(S, lower=True, overwrite_a=True), y - p.Xa.ravel()[cell])
    assert np.abs(z - zr).max() <= 2e-5 * np.abs(zr).max()            # float32 background -> float32-rounded innovation
    sel = np.random.default_rng(3).choice(p.Xa.size, 4000, replace=False)
    pg = orc.unit_vectors(p.lat.ravel()[sel], p.lon.ravel()[sel])
    inc_ref = sb[sel] * (orc.gaussian_corr(pg, po, L) @ (sb[cell] * zr))
    scale = np.abs(p.Xa).max()
    assert np.abs(inc.ravel()[sel] - inc_ref).max() <= 1e-5 * scale
    assert np.abs(xa.ravel()[sel] - (p.Xa.ravel()[sel] + inc_ref)).max() <= 1e-5 * scale


def test_dense_config3_full_size_properties(ctx):
    """BASELINE config 3 at full size (720x1440 grid, ~1e5 swath observations, S = 40 GB): no CPU solve of the whole
    system is affordable, so the result is checked through properties that do not depend on the size --
    (1) the factor is positive definite, (2) the float64 residual of (H B H^T + R) z = d is at rounding level, measured
    by the device AND re-derived on the host for random rows from the oracle's covariance formula, (3) the increment
    B H^T z at random cells equals the oracle's float64 contraction with the same z."""
    p = syn.point_obs_case(720, 1440, 100000, 3003, swaths=True)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    L = 300.0
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    m = int(y.size)
    assert m > 95000
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, y, p.obs_var)
    resid = plan.run(L, refine=2, check_pd=True, want_resid=True)          # raises if a pivot is not positive
    assert resid[-1] < 1e-7 and resid[-1] < resid[0], resid
    xa, inc = plan.download()
    z = plan.download_z()
    del plan
    assert np.isfinite(z).all() and np.isfinite(xa).all()
    sb = np.sqrt(p.Sa.ravel())
    po = orc.unit_vectors(p.obs_lat, p.obs_lon)
    d = y - p.Xa.astype(np.float32).ravel()[cell].astype(np.float64)       # the innovation the float32 plan saw
    rows = np.random.default_rng(9).choice(m, 96, replace=False)
    Srows = orc.gaussian_corr(po[rows], po, L) * sb[cell][rows][:, None] * sb[cell][None, :]
    r = d[rows] - (Srows @ z + p.obs_var[rows] * z[rows])
    assert np.abs(r).max() <= 1e-6 * np.abs(d).max(), np.abs(r).max() / np.abs(d).max()
    sel = np.random.default_rng(10).choice(p.Xa.size, 1500, replace=False)
    pg = orc.unit_vectors(p.lat.ravel()[sel], p.lon.ravel()[sel])
    inc_ref = sb[sel] * (orc.gaussian_corr(pg, po, L) @ (sb[cell] * z))
    scale = np.abs(p.Xa).max()
    assert np.abs(inc.ravel()[sel] - inc_ref).max() <= 1e-5 * scale
    assert np.abs(xa.ravel()[sel] - (p.Xa.ravel()[sel] + inc_ref)).max() <= 1e-5 * scale
    # the analysis pulls the background towards the observations
    assert np.abs(xa.ravel()[cell] - y).mean() < 0.8 * np.abs(p.Xa.ravel()[cell] - y).mean()


# ------------------------------------------------------------------------------------------------
# multi-GPU plumbing on one GPU: RCCL process group of size 1 (the N>1 logic runs on CPU/gloo in
# tests/test_parallel_cpu.py; here the device-side pieces: zero-copy view of library memory, RCCL calls)
# ------------------------------------------------------------------------------------------------
def test_rccl_plumbing_world_size_one(ctx):
    import socket
    import torch
    import torch.distributed as dist
    from oisatgmi import parallel
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        lat, lon = syn.global_grid(36, 72)
        lat2, lon2 = parallel.broadcast_grid(lat, lon, (36, 72), 0)
        np.testing.assert_array_equal(lat2, lat)
        np.testing.assert_array_equal(lon2, lon)
        p, cell = _dense_case(36, 72, 300, 1300)
        plan = dense.DenseAnalysis(lat2, lon2, max_obs=300, dtype=np.float32, ctx=ctx)
        plan.load_background(p.Xa, p.Sa)
        plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
        g = parallel.FieldGather(plan, 1, 0, 0)
        plan.run(800.0, refine=1)
        g.run()
        torch.cuda.synchronize()
        got = g.fields()
        xa, inc = plan.download()
        np.testing.assert_array_equal(got[0, 0], xa)
        np.testing.assert_array_equal(got[0, 1], inc)
        res = parallel.analyse_units(range(3), lambda u: torch.full((4,), float(u), device="cuda"),
                                     result_shape=(4,), dtype=torch.float32, device="cuda")
        assert [float(t[0]) for t in res] == [0.0, 1.0, 2.0]
    finally:
        ctx.set_stream(None)
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------
# the whole month, in the order run/job.py calls it (job.py:61-86): regrid every granule, recalculate the
# AMFs, average, bias-correct, analyse, write -- product against the oracle chained the same way
# ------------------------------------------------------------------------------------------------
def test_month_pipeline_end_to_end(ctx, tmp_path):
    import copy
    from oisatgmi.amf_recal import amf_recal
    nz_c, nz_s, ngran = 10, 8, 6
    ctm = syn.ctm_days(41, 57, nz_c, 2, 6100, averaged=False, dtype=np.float64, lat0=-20.0, lat1=20.0, lon0=-28.0, lon1=28.0)
    coord = {"Latitude": ctm[0].latitude, "Longitude": ctm[0].longitude}            # 1 deg model grid
    rng = np.random.default_rng(6101)
    swaths = []
    for g in range(ngran):
        s = syn.swath_granule(6200 + g, nscan=200, npix=50, lat0=-24.0 + g, lat1=24.0, lon_c=-12.0 + 5.0 * g, width_deg=30.0)
        s.time = s.time.replace(day=1 + g % 2, minute=10 + g)
        s.scattering_weights = rng.uniform(0.3, 2.0, size=(nz_s,) + s.vcd.shape)
        s.pressure_mid = np.linspace(1010.0, 60.0, nz_s)[:, None, None] * (1 + 0.004 * rng.normal(size=(nz_s,) + s.vcd.shape))
        s.tropopause = rng.uniform(90.0, 250.0, size=s.vcd.shape)
        swaths.append(s)

    def run(regrid, recal, avg, oi_fn, tag):
        sat = [regrid(4, 0.25, copy.deepcopy(s), coord, 0.75) for s in swaths] + [None]
        sat = recal(ctm, sat)
        r = _Reader()
        r.sat_data = sat
        return sat, avg("2019-06-01", "2019-07-01", r)

    sat_o, av_o = run(lambda *a: orc.interpolator(*a, record_type=cfg.satellite_amf), orc.amf_recal,
                      lambda a, b, r: orc.averaging(a, b, r, amf_type=cfg.satellite_amf, opt_type=cfg.satellite_opt), None, "oracle")
    sat_p, av_p = run(interpolator, amf_recal, averaging, None, "product")
    for a, b in zip(sat_p, sat_o):
        assert (a is None) == (b is None)
        if a is not None:
            for f in ("vcd", "ctm_vcd", "new_amf", "uncertainty"):
                np.testing.assert_allclose(getattr(a, f), getattr(b, f), rtol=1e-10, equal_nan=True, err_msg=f)
    for a, b in zip(av_p[:5], av_o[:5]):
        np.testing.assert_allclose(a, b, rtol=1e-10, equal_nan=True)
    assert np.isfinite(av_p[0]).sum() > 200
    o = oisatgmi()
    o.reader_obj = _Reader()
    o.reader_obj.sat_data = sat_p
    o.reader_obj.ctm_data = ctm
    o.average("2019-06-01", "2019-07-01", gasname="NO2")
    o.bias_correct("OMI", "NO2")
    y_before = o.sat_averaged_vcd.copy()
    o.oi("OMI", error_ctm=50.0)
    Xa, Y, Sa, So = orc.driver_oi_inputs(av_o[2], orc.bias_correct(av_o[0], "OMI", "NO2"), av_o[1], av_o[3], av_o[4], "OMI", 50.0)
    ref = orc.OI(Xa.copy(), Y.copy(), Sa, So, regularization_on=True)
    for a, b in zip((o.ctm_averaged_vcd_corrected, o.ak_OI, o.increment_OI, o.error_OI), ref[:4]):
        np.testing.assert_allclose(a, b, rtol=1e-9, equal_nan=True)
    path = o.write_to_nc("NO2_201906", str(tmp_path))
    assert os.path.getsize(path) > 11 * 4 * Xa.size


def test_c_abi_rejects_bad_arguments(ctx):
    """Error behaviour of the boundary: negative status + message, never a crash or a silent no-op."""
    lib = ctx.lib
    buf = ctx.alloc(1024)
    rc = lib.oisat_oi_apply(ctx.h, 7, buf.ptr, buf.ptr, buf.ptr, buf.ptr, 16, 1.0, buf.ptr, None, None, None)     # bad dtype
    assert rc == -1 and b"dtype" in lib.oisat_last_error()
    rc = lib.oisat_oi_apply(ctx.h, _hip.F32, None, buf.ptr, buf.ptr, buf.ptr, 16, 1.0, buf.ptr, None, None, None)  # NULL input
    assert rc == -1
    rc = lib.oisat_nanmean_stack(ctx.h, _hip.F32, buf.ptr, 0, 16, 0, buf.ptr)                                      # k = 0
    assert rc == -1
    rc = lib.oisat_gemm_nt(ctx.h, buf.ptr, 128, buf.ptr, 128, buf.ptr, 128, 100, 128, 128, 0, 0)                   # M not a multiple of 128
    assert rc == -1
    z = ctx.alloc(8 * 128)
    rc = lib.oisat_potrs(ctx.h, buf.ptr, 128, 128, z.ptr)                                                          # no factorization of this matrix
    assert rc == -1
    with pytest.raises(_hip.OisatError):
        ctx.check(rc)
    with pytest.raises(TypeError):
        _hip.dtype_code(np.int32)
    # a second handle on the same device is independent (own workspaces, own factor)
    other = _hip.Context(ctx.device).own_stream()
    d = other.upload(np.arange(8.0))
    np.testing.assert_array_equal(other.download(d.ptr, (8,), np.float64), np.arange(8.0))
    other.close()


def test_integration_md_stub_works_as_written(ctx, golden, tmp_path, monkeypatch):
    """INTEGRATION.md section B shows the ctypes stub a maintainer of the reference would add; run that very code block
    against the library and compare with the reference's golden OI output (forced knee index)."""
    import importlib.util
    import re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# oisatgmi/_oisat_hip\.py.*?)```", text, re.S)
    assert m, "stub code block not found in INTEGRATION.md"
    stub = tmp_path / "_oisat_hip_stub.py"
    stub.write_text(m.group(1))
    monkeypatch.setenv("OISAT_LIB", _hip.library_path())
    spec = importlib.util.spec_from_file_location("_oisat_hip_stub", str(stub))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = golden("oi_72x144.npz")
    Xa, Y, Sa, So = _oi_inputs(g)
    scales = orc.scaling_factors(True)
    Yc = Y.copy()
    Xb, AK, inc, err = mod.oi_hip(Xa, Yc, Sa, So, scales, lambda x, y: 37)
    _check_pack(g, "on37", (Xb, AK, inc, err), RT64)
    assert not (Yc[~np.isnan(Yc)] < 0).any()


def test_plain_c_client_runs_on_the_gpu(ctx, tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = _hip.library_path()
    exe = str(tmp_path / "abi_smoke")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", exe,
                    "-L", os.path.dirname(lib), "-loisat_hip", "-lm", "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"],
                   check=True)
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0 and "C ABI smoke ok" in run.stdout, run.stdout + run.stderr
