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
