"""profiling aid: a large system (default: config 3, 720x1440 / 1e5 observations) factored by the recursion and by the task graph:
time of the factorization alone (HIP events through the profile records) and the refinement residuals of the analysis.
usage: python tools/dag_big.py [M] [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ctx = _hip.context()
p = syn.point_obs_case(720, 1440, m, 4000, swaths=True)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa)
plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
lib, h = ctx.lib, ctx.h
fields = {}
for mode in os.environ.get("MODES", "0,1").split(","):
    os.environ["OISAT_DAG"] = mode
    os.environ["OISAT_DAG_MAX_BLOCKS"] = "100000"
    res = plan.run(300.0, refine=2, check_pd=True, want_resid=True)
    fields[mode] = plan.download()[0]
    ts = []
    for _ in range(reps):
        ctx.check(lib.oisat_cov_build(h, plan.oxyz.ptr, plan.osig.ptr, plan.ovar.ptr, plan.m, dense.decay_constant(300.0), plan.S.ptr, plan.mp))
        ctx.sync()
        t0 = time.perf_counter()
        ctx.check(lib.oisat_potrf(h, plan.S.ptr, plan.m, plan.mp, None))
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    plan.check()
    t = min(ts)
    print("OISAT_DAG=%s m=%d: factorization %.1f ms = %.1f TFLOP/s (%.4f of 157.3); residuals %s"
          % (mode, plan.m, t * 1e3, plan.m ** 3 / 3.0 / t / 1e12, plan.m ** 3 / 3.0 / t / 157.3e12, ["%.2e" % r for r in res]), flush=True)
if len(fields) == 2:
    a, b = fields.values()
    print("fields: max |xa_dag - xa_rec| = %.3e (scale %.3e)" % (np.abs(a - b).max(), np.abs(a).max()))
