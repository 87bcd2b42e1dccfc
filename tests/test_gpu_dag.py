"""GPU tests of the task-graph Cholesky (csrc/dense_dag.inc: one persistent launch of left-looking tile tasks) and of the
patch-wise increment.  The reference has no dense factorization (optimal_interpolation.py:27 is element-wise), so the checks
are the factorization's own: L L^T against the matrix in float64, agreement with the recursive schedule, bitwise
repeatability, the triangular solves through the factor, error reporting -- on single systems from one block row up and on
mixed batches that exercise chains, waves and the CU reservation."""
import ctypes as C
import os

import numpy as np
import pytest

from oisatgmi import _hip, dense, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _hip.context()
    assert "gfx950" in c.device_info()["name"]
    yield c
    c.check(c.lib.oisat_set_task_graph(c.h, -1))


def _system(ctx, m, seed, L_km=500.0, grid=(72, 144)):
    p = syn.point_obs_case(grid[0], grid[1], m, seed)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    oxyz = ctx.upload(dense.unit_vectors(p.obs_lat, p.obs_lon))
    osig = ctx.upload(np.sqrt(p.Sa.ravel())[cell], dtype=np.float64)
    ovar = ctx.upload(p.obs_var, dtype=np.float64)
    mp = -(-m // 128) * 128

    def build(S):
        ctx.check(ctx.lib.oisat_cov_build(ctx.h, oxyz.ptr, osig.ptr, ovar.ptr, m, dense.decay_constant(L_km), S.ptr, mp))
    return build, mp, (oxyz, osig, ovar)


def _factor(ctx, build, S, m, mp, mode):
    lib = ctx.lib
    ctx.check(lib.oisat_set_task_graph(ctx.h, mode))
    build(S)
    info = C.c_int(-1)
    ctx.check(lib.oisat_potrf(ctx.h, S.ptr, m, mp, C.byref(info)))
    assert info.value == 0
    return ctx.download(S.ptr, (mp, mp), np.float32)


@pytest.mark.parametrize("m", [128, 200, 300, 385, 640, 1000, 1664, 3000])
def test_task_graph_factor_against_the_matrix_and_the_recursion(ctx, m):
    """One system of 1 .. 24 block rows (m a multiple of 128 and not): L L^T = S to fp32 rounding in float64, the factor
    agrees with the recursive schedule's, the padding rows stay an identity, and oisat_potrs solves through it."""
    build, mp, keep = _system(ctx, m, 5000 + m)
    S = ctx.alloc(mp * mp * 4)
    build(S)
    A = ctx.download(S.ptr, (mp, mp), np.float32).astype(np.float64)
    A = np.tril(A) + np.tril(A, -1).T
    Lr = np.tril(_factor(ctx, build, S, m, mp, 0)).astype(np.float64)
    Ld = np.tril(_factor(ctx, build, S, m, mp, 1)).astype(np.float64)
    scale = np.abs(A).max()
    assert np.isfinite(Ld).all()
    assert np.abs(Ld @ Ld.T - A).max() <= 6e-7 * scale * max(1.0, np.sqrt(m / 128.0)), np.abs(Ld @ Ld.T - A).max() / scale
    assert np.abs(Ld - Lr).max() <= 2e-5 * np.abs(Lr).max()
    if mp > m:
        assert np.array_equal(Ld[m:, :m], np.zeros((mp - m, m))) and np.array_equal(Ld[m:, m:], np.eye(mp - m))
    rhs = np.random.default_rng(m).normal(size=m)
    zb = ctx.upload(rhs)
    ctx.check(ctx.lib.oisat_potrs(ctx.h, S.ptr, m, mp, zb.ptr))
    z = ctx.download(zb.ptr, (m,), np.float64)
    zr = np.linalg.solve(A[:m, :m], rhs)
    assert np.linalg.norm(z - zr) <= 1e-3 * np.linalg.norm(zr)


def test_task_graph_is_bitwise_repeatable(ctx):
    """A tile's products are accumulated in column order whatever the moment its inputs arrive (the K-loop is cut into
    segments by availability, never reordered): two runs of the same system give the same bits, and so does a run whose
    system of fewer than eight block rows, whose chain shares its CU (no reservation)."""
    for m, seed in ((2500, 77), (900, 78)):
        build, mp, keep = _system(ctx, m, seed)
        S = ctx.alloc(mp * mp * 4)
        a = _factor(ctx, build, S, m, mp, 1)
        b = _factor(ctx, build, S, m, mp, 1)
        assert np.array_equal(np.tril(a), np.tril(b))


def test_task_graph_reports_a_non_positive_pivot_and_drains(ctx):
    """A matrix that is not positive definite: the launch still drains (no workgroup waits for a block that never comes --
    the diagonal-block waves substitute the pivot and go on), the first bad column is reported, the sticky status is set."""
    lib = ctx.lib
    ctx.check(lib.oisat_set_task_graph(ctx.h, 1))
    m = 1024
    for col in (0, 130, 700, 1023):
        A = (4.0 * np.eye(m) + 0.5).astype(np.float32)
        A[col, col] = -1.0
        S = ctx.upload(A)
        info = C.c_int(-1)
        rc = lib.oisat_potrf(ctx.h, S.ptr, m, m, C.byref(info))
        assert rc != 0 and info.value == col + 1, (col, info.value)
        assert "not positive definite at column %d" % (col + 1) in lib.oisat_last_error().decode()
        ctx.solve_status(clear=True)
    A = (4.0 * np.eye(m) + 0.5).astype(np.float32)          # unchecked: the status words carry it
    A[300, 300] = -1.0
    S = ctx.upload(A)
    ctx.check(lib.oisat_potrf(ctx.h, S.ptr, m, m, None))
    col, nblk, nto = ctx.solve_status(clear=True)[:3]
    assert col == 301 and nblk >= 1 and nto == 0


@pytest.mark.parametrize("sizes", [[2100, 1500, 1290, 1000, 777, 640, 300, 257, 129, 128, 100],
                                   [2100, 640, 600, 520, 500, 480, 300, 300, 257, 257, 200, 129, 128, 100, 90, 64]])
def test_task_graph_batch_of_mixed_sizes(ctx, sizes):
    """oisat_batch_potrf as ONE task-graph launch over systems of 1 .. 17 block rows: every member's factor agrees with its own
    oisat_potrf (recursion) to fp32 rounding and reproduces its matrix; a wave 0 of five systems (the first list) and a wave 0 of
    one followed by two waves of eight and seven (the second)."""
    lib = ctx.lib
    env = {}
    try:
        mats, refs = [], []
        for k, m in enumerate(sizes):
            build, mp, keep = _system(ctx, m, 9100 + k)
            S1, S2 = ctx.alloc(mp * mp * 4), ctx.alloc(mp * mp * 4)
            refs.append(_factor(ctx, build, S1, m, mp, 0))
            build(S2)
            mats.append((S2, ctx.alloc(mp * 128 * 4), m, mp, keep))
        n = len(mats)
        Sp = (C.c_void_p * n)(*[a[0].ptr for a in mats])
        Tp = (C.c_void_p * n)(*[a[1].ptr for a in mats])
        mm = (C.c_int64 * n)(*[a[2] for a in mats])
        ld = (C.c_int64 * n)(*[a[3] for a in mats])
        bid = C.c_int(-1)
        ctx.check(lib.oisat_set_task_graph(ctx.h, 1))
        ctx.check(lib.oisat_batch_create(ctx.h, n, Sp, mm, ld, Tp, C.byref(bid)))
        for rep in range(2):                                # the second run finds the progress words handed back clean
            if rep:
                for (S2, T, m, mp, keep), _ in zip(mats, refs):
                    ctx.check(lib.oisat_cov_build(ctx.h, keep[0].ptr, keep[1].ptr, keep[2].ptr, m, dense.decay_constant(500.0), S2.ptr, mp))
            info2 = (C.c_int * 2)(-1, -1)
            ctx.check(lib.oisat_batch_potrf(ctx.h, bid.value, info2))
            assert list(info2) == [0, -1]
            for (S2, T, m, mp, keep), ref in zip(mats, refs):
                got = ctx.download(S2.ptr, (mp, mp), np.float32)
                assert np.isfinite(np.tril(got)).all()
                assert np.abs(np.tril(got) - np.tril(ref)).max() <= 2e-5 * np.abs(np.tril(ref)).max(), (m, rep)
                ctx.check(lib.oisat_factor_adopt(ctx.h, S2.ptr, m, mp, T.ptr))
                rhs = np.random.default_rng(m).normal(size=m)
                zb = ctx.upload(rhs)
                ctx.check(lib.oisat_potrs(ctx.h, S2.ptr, m, mp, zb.ptr))
                z = ctx.download(zb.ptr, (m,), np.float64)
                Lh = np.tril(ref[:m, :m]).astype(np.float64)
                zr = np.linalg.solve(Lh @ Lh.T, rhs)
                assert np.linalg.norm(z - zr) <= 1e-3 * np.linalg.norm(zr), (m, rep)
        ctx.check(lib.oisat_batch_destroy(ctx.h, bid.value))
    finally:
        for k in env:
            del os.environ[k]
        ctx.check(lib.oisat_set_task_graph(ctx.h, -1))


def test_task_graph_batch_names_the_member_that_is_not_positive_definite(ctx):
    lib = ctx.lib
    ctx.check(lib.oisat_set_task_graph(ctx.h, 1))
    bid = C.c_int(-1)
    info2 = (C.c_int * 2)(-1, -1)
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


def test_task_graph_and_recursion_give_the_same_analysis(ctx):
    """The whole dense analysis (360x720, 6,000 point observations, L = 500 km) through both schedules: fields agree to
    refinement accuracy, the float64 residuals of the gain solve meet the same tolerance."""
    p = syn.point_obs_case(360, 720, 6000, 4100)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=6000, dtype=np.float32, ctx=ctx)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
    out = {}
    for mode in (0, 1):
        ctx.check(ctx.lib.oisat_set_task_graph(ctx.h, mode))
        res = plan.run(500.0, refine=2, check_pd=True, want_resid=True)
        plan.check()
        out[mode] = (plan.download()[0], res)
    ctx.check(ctx.lib.oisat_set_task_graph(ctx.h, -1))
    scale = np.abs(p.Xa).max()
    assert np.abs(out[0][0] - out[1][0]).max() <= 2e-6 * scale
    assert out[1][1][-1] <= 1e-6 and out[0][1][-1] <= 1e-6


@pytest.mark.parametrize("shape,m", [((37, 50), 900), ((64, 96), 2000), ((120, 1440), 3000)])
def test_increment_by_patches_equals_the_increment_by_runs(ctx, shape, m):
    """oisat_apply_increment_grid (32-wide patches of the ny x nx grid, observations beyond the covariance's reach of a patch
    skipped) against oisat_apply_increment (runs of consecutive cells): the same sums up to terms below 2^-64 of a term and
    their order -- on grids whose edges cut the patches, and on a polar band (120 x 1440 cells at 60-90 degrees north)
    where most of the latitude window lies beyond the pole."""
    lib = ctx.lib
    ny, nx = shape
    rng = np.random.default_rng(m)
    if nx == 1440:
        lat = np.linspace(60.125, 89.875, ny)[:, None] * np.ones((1, nx))
        lon = np.ones((ny, 1)) * np.linspace(-179.875, 179.875, nx)[None, :]
        olat, olon = rng.uniform(52.0, 90.0, m), rng.uniform(-180.0, 180.0, m)
    else:
        lat = np.linspace(-20.0, 25.0, ny)[:, None] * np.ones((1, nx))
        lon = np.ones((ny, 1)) * np.linspace(100.0, 160.0, nx)[None, :]
        olat, olon = rng.uniform(-28.0, 33.0, m), rng.uniform(92.0, 168.0, m)
    order = np.argsort(olat, kind="stable")
    olat, olon = olat[order], olon[order]
    n = ny * nx
    gxyz = ctx.upload(dense.unit_vectors(lat.ravel(), lon.ravel()))
    gsig = ctx.upload(rng.uniform(0.5, 1.5, n))
    glat = ctx.upload(lat.ravel().astype(np.float64))
    oxyz = ctx.upload(dense.unit_vectors(olat, olon))
    osig = ctx.upload(rng.uniform(0.5, 1.5, m))
    z = ctx.upload(rng.normal(size=m))
    olat_d = ctx.upload(olat.astype(np.float64))
    xb = ctx.upload(rng.normal(size=n).astype(np.float32))
    g = dense.decay_constant(300.0)
    outs = []
    for grid in (False, True):
        xa, inc = ctx.alloc(n * 4), ctx.alloc(n * 4)
        if grid:
            ctx.check(lib.oisat_apply_increment_grid(ctx.h, 0, gxyz.ptr, gsig.ptr, ny, nx, oxyz.ptr, osig.ptr, z.ptr, m, g, xb.ptr, xa.ptr,
                                                     inc.ptr, glat.ptr, olat_d.ptr))
        else:
            ctx.check(lib.oisat_apply_increment(ctx.h, 0, gxyz.ptr, gsig.ptr, n, oxyz.ptr, osig.ptr, z.ptr, m, g, xb.ptr, xa.ptr, inc.ptr,
                                                glat.ptr, olat_d.ptr))
        outs.append((ctx.download(xa.ptr, (n,), np.float32), ctx.download(inc.ptr, (n,), np.float32)))
    (xa0, inc0), (xa1, inc1) = outs
    assert np.isfinite(inc1).all() and np.abs(inc0).max() > 0
    assert np.abs(inc1 - inc0).max() <= 2e-7 * np.abs(inc0).max()
    assert np.abs(xa1 - xa0).max() <= 2e-7 * max(np.abs(xa0).max(), 1.0)
    # and against the float64 contraction on a sample of cells
    sel = rng.choice(n, 200, replace=False)
    pg, po = dense.unit_vectors(lat.ravel()[sel], lon.ravel()[sel]), dense.unit_vectors(olat, olon)
    pg, po = np.asarray(pg).reshape(3, -1), np.asarray(po).reshape(3, -1)
    d2 = ((pg[:, :, None] - po[:, None, :]) ** 2).sum(axis=0)
    w = ctx.download(osig.ptr, (m,), np.float64) * ctx.download(z.ptr, (m,), np.float64)
    ref = ctx.download(gsig.ptr, (n,), np.float64)[sel] * (np.exp(-g * d2) @ w)
    assert np.abs(inc1[sel] - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)


@pytest.mark.parametrize("m,region", [(900, "box"), (3000, "cap"), (5000, "globe")])
def test_residual_on_compact_blocks_equals_the_residual_by_latitude_rows(ctx, m, region):
    """oisat_cov_residual with blocks of rows taken along the Morton curve (oisat_set_obs_blocks: bounding sphere, distance
    cull of the latitude window's observations) against the plain form (64 consecutive latitudes per block) and against the
    float64 contraction: the same terms per row in the same order up to terms below 2^-64 of a term."""
    lib = ctx.lib
    rng = np.random.default_rng(m)
    if region == "box":
        lat, lon = rng.uniform(-28.0, 33.0, m), rng.uniform(92.0, 168.0, m)
    elif region == "cap":
        lat, lon = rng.uniform(52.0, 90.0, m), rng.uniform(-180.0, 180.0, m)
    else:
        lat, lon = np.degrees(np.arcsin(rng.uniform(-1.0, 1.0, m))), rng.uniform(-180.0, 180.0, m)
    o = np.argsort(lat, kind="stable")
    lat, lon = lat[o], lon[o]
    po = dense.unit_vectors(lat, lon)
    oxyz = ctx.upload(po)
    sig, var = rng.uniform(0.5, 1.5, m), rng.uniform(0.1, 0.3, m)
    zz, dd = rng.normal(size=m), rng.normal(size=m)
    osig, ovar, z, d = ctx.upload(sig), ctx.upload(var), ctx.upload(zz), ctx.upload(dd)
    olat = ctx.upload(lat.astype(np.float64))
    perm = ctx.upload(dense.morton_order(lat, lon))
    g = dense.decay_constant(300.0)
    outs = []
    for blocks in (False, True):
        r = ctx.alloc(m * 8)
        ctx.check(lib.oisat_set_obs_blocks(ctx.h, perm.ptr if blocks else None, m if blocks else 0))
        ctx.check(lib.oisat_cov_residual(ctx.h, oxyz.ptr, osig.ptr, ovar.ptr, m, g, d.ptr, z.ptr, r.ptr, olat.ptr))
        outs.append(ctx.download(r.ptr, (m,), np.float64))
    ctx.check(lib.oisat_set_obs_blocks(ctx.h, None, 0))
    d2 = ((po[:, :, None] - po[:, None, :]) ** 2).sum(axis=0)
    ref = dd - (sig * ((np.exp(-g * d2) * sig[None, :]) @ zz) + var * zz)
    scale = np.abs(ref).max()
    assert np.abs(outs[1] - outs[0]).max() <= 1e-12 * scale
    assert np.abs(outs[1] - ref).max() <= 1e-11 * scale


def test_task_graph_under_concurrent_uneven_load_is_bitwise_stable(ctx):
    """Three systems factored over and over on three streams (three task-graph launches sharing the CUs, workgroups starting
    late, chains losing their reserved CU) next to bursts of memory traffic on a fourth: every factor equals the first one of
    its system bit for bit.  A hand-over that reads a stale L1 or L2 line shows here (tools/dag_stress.py is the long form:
    the missing L1 drop behind SUB(j) showed as one wrong factor in ~5,000)."""
    import threading
    import time
    dev = ctx.device
    stop = []
    bad, counts = [], {}

    def worker(tag, m, seed):
        c = _hip.Context(dev).own_stream()
        c.bind_thread()
        build, mp, keep = _system(c, m, seed, grid=(180, 360))
        S = c.alloc(mp * mp * 4)
        c.check(c.lib.oisat_set_task_graph(c.h, 1))

        def run():
            build(S)
            c.check(c.lib.oisat_potrf(c.h, S.ptr, m, mp, None))
        run()
        c.sync()
        ref = np.tril(c.download(S.ptr, (mp, mp), np.float32)).view(np.uint32).copy()
        n = 0
        while not stop:
            for _ in range(5):
                run()
            c.sync()
            got = np.tril(c.download(S.ptr, (mp, mp), np.float32)).view(np.uint32)
            if not np.array_equal(got, ref):
                bad.append((tag, n, int((got != ref).sum())))
            n += 5
        counts[tag] = n
        if not c.solve_status(clear=True).clean:
            bad.append((tag, "status"))
        c.close()

    def noise():
        c = _hip.Context(dev).own_stream()
        c.bind_thread()
        b = c.alloc(256 << 20)
        n = 0
        while not stop:
            for _ in range(20):
                c.check(c.lib.oisat_memset(c.h, b.ptr, n & 255, b.nbytes))
            c.sync()
            time.sleep(0.003 * (n % 3))
            n += 1
        c.close()

    threads = [threading.Thread(target=worker, args=("a", 2500, 11)), threading.Thread(target=worker, args=("b", 4100, 12)),
               threading.Thread(target=worker, args=("c", 900, 13)), threading.Thread(target=noise)]
    for th in threads:
        th.start()
    time.sleep(12.0)
    stop.append(1)
    for th in threads:
        th.join()
    assert not bad, bad
    assert min(counts.values()) >= 50, counts


def test_task_graph_time_out_drains_the_launch_and_is_reported(ctx):
    """Fault injection -- only the -DOISAT_TEST_HOOKS build of the library has it (liboisat_hip_testhooks.so; OISAT_DAG_FLAGS =
    128 | 256 there: the chains stop announcing their diagonal blocks at block 3, polls give up after 4 096 rounds): the waiting
    tile tasks time out, raise the launch's error word, every other workgroup sees it and leaves -- the launch ends within
    milliseconds instead of hanging --, the time-out is reported as a TASK-GRAPH time-out (its own status word; the checked call
    returns an error that names the factorization), and the NEXT factorization on the same plan finds the progress words clean
    and is correct.  The product library ignores the variable."""
    hooks = _hip.Context(ctx.device, lib=_hip.load_test_hooks_library()).own_stream()
    lib = hooks.lib
    m = 1500
    build, mp, keep = _system(hooks, m, 4242)
    S = hooks.alloc(mp * mp * 4)
    good = _factor(hooks, build, S, m, mp, 1)
    hooks.solve_status(clear=True)
    os.environ["OISAT_DAG_FLAGS"] = str(128 | 256)
    try:
        build(S)
        hooks.check(lib.oisat_set_task_graph(hooks.h, 1))
        hooks.check(lib.oisat_potrf(hooks.h, S.ptr, m, mp, None))
        hooks.sync()
        st = hooks.solve_status(clear=True)
        assert st.dag_timeouts >= 1 and st.trsv_timeouts == 0 and st.notpd_col == 0 and not st.clean
        build(S)
        info = C.c_int(-1)
        rc = lib.oisat_potrf(hooks.h, S.ptr, m, mp, C.byref(info))          # checked: the call itself says what happened
        assert rc != 0 and "task-graph factorization timed out" in lib.oisat_last_error().decode()
        assert hooks.solve_status(clear=True).clean                          # reported once
        with pytest.raises(_hip.OisatError, match="task-graph factorization"):
            build(S)
            hooks.check(lib.oisat_potrf(hooks.h, S.ptr, m, mp, None))
            hooks.check_solves()
        # the product library does not know the variable
        build2, mp2, keep2 = _system(ctx, m, 4242)
        S2 = ctx.alloc(mp2 * mp2 * 4)
        assert np.array_equal(np.tril(_factor(ctx, build2, S2, m, mp2, 1)), np.tril(good))
        assert ctx.solve_status(clear=True).clean
    finally:
        del os.environ["OISAT_DAG_FLAGS"]
    again = _factor(hooks, build, S, m, mp, 1)
    assert np.array_equal(np.tril(again), np.tril(good))
    assert hooks.solve_status(clear=True).clean
    hooks.close()
    ctx.check(ctx.lib.oisat_set_task_graph(ctx.h, -1))
