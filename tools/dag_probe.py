"""profiling aid: ONE system factored by the recursive schedule and by the task graph (OISAT_DAG=0 | 1) -- factors compared
(L L^T against S in float64 on a sample of block rows, the two L against each other), analysis fields compared, time per
analysis and per factorization.
usage: python tools/dag_probe.py M [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import ctypes as C
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = _hip.context()
ny, nx = (360, 720) if m <= 20000 else (720, 1440)
p = syn.point_obs_case(ny, nx, m, 4000, swaths=m > 20000)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa)
plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
L_km = 500.0 if m <= 20000 else 300.0
lib, h = ctx.lib, ctx.h


def factor_only():
    ctx.check(lib.oisat_cov_build(h, plan.oxyz.ptr, plan.osig.ptr, plan.ovar.ptr, plan.m, dense.decay_constant(L_km), plan.S.ptr, plan.mp))
    ctx.check(lib.oisat_potrf(h, plan.S.ptr, plan.m, plan.mp, None))


out = {}
for mode in ("0", "1"):
    os.environ["OISAT_DAG"] = mode
    res = plan.run(L_km, refine=2, check_pd=True, want_resid=True)
    plan.check()
    xa, inc = plan.download()
    Lf = np.tril(plan.download_S()[:plan.m, :plan.m]).astype(np.float64)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        plan.run(L_km, refine=2)
    ctx.sync()
    t_run = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        factor_only()
    ctx.sync()
    t_fac = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.check(lib.oisat_cov_build(h, plan.oxyz.ptr, plan.osig.ptr, plan.ovar.ptr, plan.m, dense.decay_constant(L_km), plan.S.ptr, plan.mp))
    ctx.sync()
    t_build = (time.perf_counter() - t0) / reps
    plan.check()
    out[mode] = (xa, inc, Lf, res)
    fl = plan.m ** 3 / 3.0
    print("OISAT_DAG=%s m=%d: analysis %.3f ms, build+factor %.3f ms (build %.3f) -> factor %.3f ms = %.1f TFLOP/s; residuals %s"
          % (mode, plan.m, t_run * 1e3, t_fac * 1e3, t_build * 1e3, (t_fac - t_build) * 1e3, fl / (t_fac - t_build) / 1e12,
             ["%.2e" % r for r in res]), flush=True)

# the factor against the matrix: rebuild S on the host for a sample of rows (float64) and compare with L L^T
ctx.check(lib.oisat_cov_build(h, plan.oxyz.ptr, plan.osig.ptr, plan.ovar.ptr, plan.m, dense.decay_constant(L_km), plan.S.ptr, plan.mp))
S = plan.download_S()[:plan.m, :plan.m].astype(np.float64)
S = np.tril(S) + np.tril(S, -1).T
rows = np.unique(np.concatenate([np.arange(0, min(plan.m, 300)), np.random.default_rng(0).integers(0, plan.m, 600), np.arange(plan.m - 300, plan.m)]))
for mode in ("0", "1"):
    Lf = out[mode][2]
    R = Lf[rows] @ Lf.T - S[rows]
    print("OISAT_DAG=%s: max |L L^T - S| on %d rows = %.3e (max |S| = %.3e)" % (mode, len(rows), np.abs(R).max(), np.abs(S).max()))
d = np.abs(out["0"][2] - out["1"][2])
print("max |L_dag - L_rec| = %.3e at %s; fields: max |xa_dag - xa_rec| = %.3e (scale %.3e)"
      % (d.max(), np.unravel_index(np.argmax(d), d.shape), np.abs(out["0"][0] - out["1"][0]).max(), np.abs(out["0"][0]).max()))
