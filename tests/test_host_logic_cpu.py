"""Host-side logic of the drop-in package that needs no GPU."""
import dataclasses
import datetime
import os

import numpy as np
import pytest

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


def test_oi_mode_settings_are_read_from_attributes_then_environment(monkeypatch):
    """The analysis-mode switches of oisatgmi.oi() (driver.py:108-114 keeps its signature): an attribute on the instance
    wins over the environment, which wins over the default; a bad value is refused before anything touches the GPU."""
    import pytest
    o = oisatgmi()
    assert o._oi_setting("oi_mode", "OISAT_OI_MODE", "diag") == "diag"
    monkeypatch.setenv("OISAT_OI_MODE", "tiled")
    monkeypatch.setenv("OISAT_CORR_LENGTH_KM", "450")
    assert o._oi_setting("oi_mode", "OISAT_OI_MODE", "diag") == "tiled"
    assert o._oi_setting("corr_length_km", "OISAT_CORR_LENGTH_KM", 300.0, float) == 450.0
    o.oi_mode, o.corr_length_km = "dense", 120
    assert o._oi_setting("oi_mode", "OISAT_OI_MODE", "diag") == "dense"
    assert o._oi_setting("corr_length_km", "OISAT_CORR_LENGTH_KM", 300.0, float) == 120.0
    o.ctm_averaged_vcd = np.ones((4, 8))
    o.sat_averaged_vcd = np.ones((4, 8))
    o.sat_averaged_error = np.ones((4, 8))
    o.oi_mode = "banana"
    with pytest.raises(ValueError, match="diag, dense or tiled"):
        o.oi("OMI")
    # the model grid: explicit attributes, else the first granule's centres (as output_fields does, driver.py:190-195)
    lat, lon = syn.global_grid(4, 8)
    o.grid_lat, o.grid_lon = lat, lon
    np.testing.assert_array_equal(o._oi_grid()[0], lat)
    o2 = oisatgmi()

    class R:
        pass
    o2.reader_obj = R()
    o2.reader_obj.sat_data = [None, syn.granule_stack(4, 8, 2, 1)[0]]
    np.testing.assert_array_equal(o2._oi_grid()[1], o2.reader_obj.sat_data[1].longitude_center)


def test_tile_partition_polar_caps_and_halo():
    """dense.tile_partition: every cell in exactly one tile; a band whose halo reaches the pole is ONE cap tile holding every
    observation of the band (+ halo); cutting it by longitude instead gives tiles with that same observation set each."""
    lat, lon = syn.global_grid(36, 72)
    rng = np.random.default_rng(3)
    olat, olon = rng.uniform(-89, 89, 3000), rng.uniform(-180, 180, 3000)
    merged = dense.tile_partition(lat, lon, olat, olon, 30.0, 900.0)
    cut = dense.tile_partition(lat, lon, olat, olon, 30.0, 900.0, merge_polar=False)
    assert len(cut) == 6 * 12 and len(merged) == 1 + 4 * 12 + 1
    cover = np.zeros((36, 72), dtype=int)
    for t in merged:
        cover[t["rows"][0]:t["rows"][1], t["cols"][0]:t["cols"][1]] += 1
    assert (cover == 1).all()
    south = merged[0]
    assert south["cols"] == (0, 72) and south["rows"] == (0, 6)
    for t in cut[:12]:
        np.testing.assert_array_equal(np.sort(t["obs"]), np.sort(south["obs"]))
    h = np.rad2deg(900.0 / dense.EARTH_RADIUS_KM)
    assert set(south["obs"]) == set(np.flatnonzero(olat <= -60.0 + h))
    # a mid-latitude tile: everything inside it is there, nothing farther than the halo (in latitude) is
    t = merged[1 + 12 + 5]
    (y0, y1), (x0, x1) = t["rows"], t["cols"]
    la0, la1 = lat[y0, 0] - 2.5, lat[y1 - 1, 0] + 2.5
    lo0, lo1 = lon[0, x0] - 2.5, lon[0, x1 - 1] + 2.5
    inside = (olat >= la0) & (olat <= la1) & (olon >= lo0) & (olon <= lo1)
    assert np.isin(np.flatnonzero(inside), t["obs"]).all()
    assert (olat[t["obs"]] >= la0 - h - 1e-9).all() and (olat[t["obs"]] <= la1 + h + 1e-9).all()


# ------------------------------------------------------------------------------------------------
# task-graph factorization: the ticket order (csrc/dense_dag.inc), checked on the host
# ------------------------------------------------------------------------------------------------
def _dag_order(block_rows, wave=0):
    import ctypes as C
    from oisatgmi import _hip
    if not os.path.exists(_hip.library_path()):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(_hip.library_path())
    lib.oisat_dag_task_order.restype = C.c_int
    lib.oisat_dag_task_order.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                         C.POINTER(C.c_int)]
    nb = np.asarray(block_rows, dtype=np.int32)
    n, res, mwc = C.c_int64(0), C.c_int(0), C.c_int(0)
    assert lib.oisat_dag_task_order(len(nb), nb.ctypes.data, wave, None, 0, C.byref(n), C.byref(res), C.byref(mwc)) == 0
    out = np.zeros((n.value, 4), dtype=np.int32)
    assert lib.oisat_dag_task_order(len(nb), nb.ctypes.data, wave, out.ctypes.data, n.value, C.byref(n), C.byref(res), C.byref(mwc)) == 0
    return out, res.value, mwc.value


MONTH = sorted([137, 137] + [48, 47, 45, 44, 40, 33, 31] * 6 + [46] * 6, reverse=True)       # block rows of a localised 720x1440 month


@pytest.mark.parametrize("block_rows,wave", [([79], 0), (MONTH, 8), ([17, 12, 11, 8, 7, 5, 3, 3, 2, 1, 1], 2), ([20, 9, 9, 8, 8, 7, 3, 2, 1], 5)])
def test_task_graph_ticket_order_is_topological_and_complete(block_rows, wave):
    """Every tile of every system is owned by exactly one task, and every input of a task carries a LOWER ticket or is a step of
    its system's chain, whose own ticket is lower (csrc/dense_dag.inc).  Inputs of a task on tile (i, j) of system s: the final
    blocks L(i, k), L(j, k), k < j (k < j - 1 for PRE), each produced by the tile task T(., k) of that row -- or by the chain for
    the sub-diagonal tile -- and the diagonal block j (chain)."""
    tasks, reserve, _ = _dag_order(block_rows, wave)
    ticket = {}
    chain_ticket = {}
    for t, (kind, s, i, j) in enumerate(tasks.tolist()):
        if kind == 0:
            assert s not in chain_ticket
            chain_ticket[s] = t
        else:
            assert (s, i, j) not in ticket
            ticket[(s, i, j)] = (t, kind)
    assert sorted(chain_ticket) == list(range(len(block_rows)))
    for s, nb in enumerate(block_rows):
        owned = {(i, j) for (q, i, j) in ticket if q == s}
        expect = {(i, j) for j in range(nb) for i in range(j + 2, nb)} | {(j + 1, j) for j in range(1, nb - 1)} | {(j, j) for j in range(2, nb)}
        assert owned == expect, s
        for (q, i, j), (t, kind) in ticket.items():
            if q != s:
                continue
            assert chain_ticket[s] < t
            assert kind == (3 if i == j else 2 if i == j + 1 else 1)
            for k in range(j - 1 if kind == 3 else j):          # producers of L(i, k) and L(j, k)
                for r in {i, j}:
                    if r >= k + 2:
                        assert ticket[(s, r, k)][0] < t, (s, i, j, r, k)       # a tile task of a lower column
                    # r == k + 1: the chain's panel tile; r == k cannot happen (k < j <= i)
    # the chains that get a CU to themselves hold the first tickets
    assert all(tasks[t][0] == 0 for t in range(reserve))


def test_task_graph_chains_must_leave_room_for_the_tasks_they_wait_for():
    """a chain waits for sub(j) / pre(j+1), which come from HIGHER tickets -- if every resident workgroup held a chain
    nobody would draw them.  The library counts the chains that can be resident at one time (the largest wave's plus those of
    one more wave behind wave 0: a wave's chains still walk their last columns when the next wave's are drawn) and only
    launches a task graph on four times as many workgroups; this is that count."""
    assert _dag_order([79])[2] == 1
    assert _dag_order(MONTH)[2] == 8 + 8                               # eight tiles at a time, the two polar caps as a pair among them
    assert _dag_order([40] * 100)[2] == 8 + 8                          # 64 "big" ones in pairs spread through the waves of the other 36
    assert _dag_order([40] * 30, wave=3)[2] == 30                      # all of them within a factor two and nothing smaller: one wave
    assert _dag_order([40, 40] + [10] * 30, wave=20)[2] == 20 + 10
    # a pair of big systems opens the launch, the others are spread evenly through the small waves (twelve months: 24 caps)
    tasks, reserve, _ = _dag_order(MONTH, 8)
    chain_sys = [int(s) for (kind, s, a, b) in tasks.tolist() if kind == 0]
    assert chain_sys[:2] == [0, 1] and chain_sys[-1] >= 2
    twelve = sorted(MONTH * 12, reverse=True)
    tasks, reserve, mwc = _dag_order(twelve, 8)
    chain_sys = [int(s) for (kind, s, a, b) in tasks.tolist() if kind == 0]
    pos = [chain_sys.index(b) for b in range(0, 24, 2)]
    assert pos[0] == 0 and all(40 <= b - a <= 60 for a, b in zip(pos, pos[1:])) and mwc == 16


def _run_bench(*argv, env_extra=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=600)


def test_bench_launches_its_own_ranks_from_a_plain_shell():
    """`python bench.py --gpus N` with no launcher around it: N child ranks under torch.distributed.run,
    arguments passed through, exactly ONE JSON line on stdout, the ranks counted by the backend itself."""
    import json
    import bench
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "9"], 2345)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "2345"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "9"] and cmd[-5].endswith("bench.py")
    r = _run_bench("--gpus", "2", "--steps", "7", "--warmup", "2", "--launcher-selftest", "0")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line == {"launcher_selftest": True, "world_size": 2, "n_ranks_seen_by_backend": 2, "gpus": 2, "steps": 7, "warmup": 2,
                    "self_launched": True}


def test_bench_launcher_propagates_a_failing_rank_and_refuses_a_mismatched_launch():
    r = _run_bench("--gpus", "2", "--launcher-selftest", "3")           # rank 3 % 2 = 1 exits with 3
    assert r.returncode != 0
    # under a launcher that started another number of ranks than --gpus says: refused before anything touches a GPU
    r = _run_bench("--gpus", "2", "--launcher-selftest", "0", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_triangulation_worker_protocol_tells_a_failure_from_a_death():
    """interpolator_many's child processes (oisatgmi/_qhull_worker.py): a triangulation comes back with its lazily computed
    members, points qhull refuses come back as ``None`` (the reference skips such a granule, interpolator.py:151-155), stray
    prints do not reach the reply pipe, and a worker that is gone raises instead of reading as "qhull failed"."""
    import subprocess
    import sys
    from oisatgmi import _qhull_worker as w
    from oisatgmi.interpolator import _QhullWorkers
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(0)
    lon, lat = rng.uniform(size=400), rng.uniform(size=400)
    procs = _QhullWorkers(2)
    try:
        with ThreadPoolExecutor(max_workers=2) as ex:
            tri = procs.submit(ex, lon, lat).result(timeout=120)
            flat = procs.submit(ex, np.arange(12.0), np.zeros(12)).result(timeout=120)
            from scipy.spatial import Delaunay
            want = Delaunay(np.column_stack((lon, lat)))
            assert np.array_equal(tri.simplices, want.simplices) and np.array_equal(tri.transform, want.transform, equal_nan=True)
            assert flat is None
            for p in procs.procs:
                p.kill()
                p.wait()
            with pytest.raises(RuntimeError, match="triangulation worker .* died"):
                procs.submit(ex, lon, lat).result(timeout=120)
    finally:
        procs.close()
