"""profiling aid: per-kernel HIP-event times of the gain solve of one system (sweeps run vs skipped after convergence)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
ctx = _hip.context()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 17500
p = syn.point_obs_case(360, 720, m, 99, swaths=True)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa)
plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
print("resid", plan.run(300.0, refine=2, check_pd=True, want_resid=True))
for tol in (None, 0.0):
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(5):
        plan.run(300.0, refine=2, tol=tol)
    prof = ctx.prof_collect(); ctx.prof_enable(False)
    print("tol", tol, {k: (v["launches"], round(v["total_ms"], 3)) for k, v in prof.items() if k.startswith(("trsv", "cov_res", "resid", "copy"))})
