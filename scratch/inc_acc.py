import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
from oracle import oi_oracle as orc
ctx = _hip.context()
for (ny, nx, m, L, sw) in ((360, 720, 10000, 500.0, False), (720, 1440, 100000, 300.0, True)):
    p = syn.point_obs_case(ny, nx, m, 3003, swaths=sw)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    y = np.where(p.obs_y < 0, 0, p.obs_y)
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=int(y.size), dtype=np.float32)
    plan.load_background(p.Xa, p.Sa); plan.load_obs(p.obs_lat, p.obs_lon, cell, y, p.obs_var)
    res = plan.run(L, refine=2, check_pd=True, want_resid=True)
    xa, inc = plan.download(); z = plan.download_z()
    ctx.prof_reset(); ctx.prof_enable(True); plan.run(L, refine=1); pr = ctx.prof_collect(); ctx.prof_enable(False)
    del plan
    sb = np.sqrt(p.Sa.ravel()); po = orc.unit_vectors(p.obs_lat, p.obs_lon)
    sel = np.random.default_rng(10).choice(p.Xa.size, 1500, replace=False)
    pg = orc.unit_vectors(p.lat.ravel()[sel], p.lon.ravel()[sel])
    C = orc.gaussian_corr(pg, po, L)
    inc_ref = sb[sel] * (C @ (sb[cell] * z))
    absum = sb[sel] * (np.abs(C) @ np.abs(sb[cell] * z))
    e = np.abs(inc.ravel()[sel] - inc_ref)
    print(ny, nx, m, "resid", res, "max err/scale %.3e" % (e.max() / np.abs(p.Xa).max()), "cancellation kappa med %.1f max %.1f" % (np.median(absum / np.maximum(np.abs(inc_ref), 1e-30)), (absum / np.abs(p.Xa).max()).max()),
          "apply_increment %.3f ms" % pr["apply_increment"]["total_ms"])
