"""Host-side logic of the drop-in package that needs no GPU."""
import dataclasses
import datetime

import numpy as np

from oisatgmi import config as cfg
from oisatgmi import synthetic as syn
from oisatgmi._kneedle import knee_index
from oisatgmi.optimal_interpolation import scaling_factors
from oisatgmi.driver import BIAS_CORRECTIONS, O3_DIVISOR, oisatgmi
from oisatgmi import interpolator as itp
from oisatgmi import dense
from oracle import oi_oracle as orc


def test_scaling_factors_are_the_reference_sweep(golden):
    g = golden("oi_72x144.npz")
    np.testing.assert_array_equal(scaling_factors(True), g["curve_x"])
    np.testing.assert_array_equal(scaling_factors(False), [1.0])
    assert scaling_factors(True).size == 99


def test_product_kneedle_agrees_with_oracle_kneedle(golden):
    """Two independent restatements of kneed's algorithm (kneed itself: parity unpinned)."""
    rng = np.random.default_rng(11)
    x = np.arange(0.1, 10, 0.1)
    for t in range(300):
        kind = t % 4
        if kind == 0:
            y = x / (x + rng.uniform(0.05, 20))
        elif kind == 1:
            y = np.cumsum(rng.uniform(0, 1, size=x.size))
        elif kind == 2:
            y = rng.normal(size=x.size)
        else:
            y = 1 - np.exp(-x * rng.uniform(0.01, 3))
        assert knee_index(x, y) == orc.kneedle_knee(x, y)[1]
    for tag in ("oi_72x144", "oi_360x720", "oi_o3_72x144"):
        g = golden(tag + ".npz")
        assert knee_index(g["curve_x"], g["curve_y"]) == orc.kneedle_knee(g["curve_x"], g["curve_y"])[1]
    assert knee_index(x, 2 * x + 1) is None and knee_index(x, np.full_like(x, np.nan)) is None
    assert knee_index(x, 3 * x) == orc.kneedle_knee(x, 3 * x)[1]       # rounding noise on a line: same pick


def test_records_are_positional_and_named_like_the_reference(golden):
    g = golden("records.npz")
    for nm in ("satellite_amf", "satellite_opt", "satellite_ssmis", "ctm_model"):
        assert [f.name for f in dataclasses.fields(getattr(cfg, nm))] == list(g[nm])
    r = cfg.satellite_ssmis(1, 2, datetime.datetime(2020, 1, 1), 4, 5, False, 7, "SSMIS")
    assert (r.vcd, r.uncertainty, r.sensor) == (1, 2, "SSMIS")
    import pickle
    assert pickle.loads(pickle.dumps(r)) == r              # crosses joblib workers like the reference's


def test_bias_table_and_o3_divisor():
    assert BIAS_CORRECTIONS == orc.BIAS_TABLE
    assert O3_DIVISOR == 2.69e16 * 1e-15
    o = oisatgmi()
    o.sat_averaged_vcd = np.array([1.0, 2.0])
    o.bias_correct("OMI", "NO2")
    np.testing.assert_array_equal(o.sat_averaged_vcd, (np.array([1.0, 2.0]) - 0.32) / 0.63)
    o.bias_correct("OMPS", "O3")                            # unknown pair: unchanged
    np.testing.assert_array_equal(o.sat_averaged_vcd, (np.array([1.0, 2.0]) - 0.32) / 0.63)


def test_box_kernels(golden):
    g = golden("upscaler.npz")
    np.testing.assert_array_equal(itp._boxfilter(3, 4), g["box_3_4"])
    np.testing.assert_array_equal(itp._boxfilter2(3, 4), g["box2_3_4"])


def test_dense_helpers():
    lat, lon = syn.global_grid(18, 36)
    p = dense.unit_vectors(lat, lon)
    assert p.shape == (3, 18 * 36)
    np.testing.assert_allclose((p * p).sum(axis=0), 1.0, rtol=1e-15)
    np.testing.assert_allclose(p.T, orc.unit_vectors(lat.ravel(), lon.ravel()), rtol=0, atol=0)
    cells = dense.regular_grid_cell(lat, lon, lat.ravel(), lon.ravel())
    np.testing.assert_array_equal(cells, np.arange(lat.size))
    assert dense.decay_constant(500.0) == 0.5 * (6371.0 / 500.0) ** 2
    # Gaussian in chord distance is positive definite on the sphere
    C = orc.gaussian_corr(p.T[::7], p.T[::7], 800.0)
    assert np.linalg.eigvalsh(C).min() > -1e-10


def test_synthetic_is_seeded():
    a = syn.diag_case(12, 24, 50, 9)
    b = syn.diag_case(12, 24, 50, 9)
    np.testing.assert_array_equal(a.Y, b.Y)
    assert (a.Y[~np.isnan(a.Y)] < 0).sum() >= 0
    p = syn.point_obs_case(36, 72, 300, 3, swaths=True)
    assert p.obs_y.size <= 300 and np.all(np.abs(p.obs_lat) <= 90)


def test_savedaily_writes_the_reference_files(tmp_path):
    """driver.py:135-155: one .mat per granule, named and keyed as the reference does (host-only, no GPU)."""
    import datetime
    from scipy.io import loadmat
    from oisatgmi.driver import oisatgmi
    from oisatgmi import config as cfg
    lat, lon = np.meshgrid(np.arange(3.0), np.arange(4.0), indexing="ij")
    mk = lambda d: cfg.satellite_ssmis(np.full((3, 4), float(d)), np.ones((3, 4)), datetime.datetime(2019, 6, d, 12), lat, lon,
                                       False, np.full((3, 4), 2.0 * d), "SSMIS")   # noqa: E731
    o = oisatgmi()
    o.reader_obj = type("RO", (), {})()
    o.reader_obj.sat_data = [mk(1), None, mk(2)]
    o.reader_obj.ctm_data = [cfg.ctm_model(lat, lon, [], [], [], [], [], "FREE", False)] * 3
    o.savedaily(str(tmp_path / "daily"), "H2O", "2019_06")
    files = sorted(p.name for p in (tmp_path / "daily").iterdir())
    assert files == ["sat_data_H2O_20190601.50.mat", "sat_data_H2O_20190602.52.mat"]
    m = loadmat(str(tmp_path / "daily" / files[1]))
    assert set(("vcd_sat", "vcd_ctm", "vcd_err", "time_sat", "lat", "lon")) <= set(m)
    np.testing.assert_array_equal(m["vcd_ctm"], np.full((3, 4), 4.0))
