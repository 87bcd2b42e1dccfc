import sys, traceback
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
from oracle import oi_oracle as orc
for m in (0, 1, 2, 5, 127, 128, 129, 257):
    p = syn.point_obs_case(36, 72, max(m, 1), 77 + m)
    sl = slice(0, m)
    obs = dict(lat=p.obs_lat[sl], lon=p.obs_lon[sl], y=p.obs_y[sl], var=p.obs_var[sl])
    try:
        xb, inc, info = dense.OI_dense(p.Xa, None, p.Sa, None, p.lat, p.lon, 600.0, refine=2, dtype=np.float32, obs=obs, want_error=(m > 0))
        if m:
            cell = dense.regular_grid_cell(p.lat, p.lon, obs["lat"], obs["lon"])
            ref = orc.dense_oi(p.lat, p.lon, p.Xa, p.Sa, obs["lat"], obs["lon"], cell, np.where(obs["y"] < 0, 0, obs["y"]), obs["var"], 600.0)
            print(m, "err/scale %.2e" % (np.abs(xb.ravel() - ref["xa"]).max() / np.abs(ref["xa"]).max()), info["residuals"])
        else:
            print(m, "ok", np.abs(inc).max(), np.array_equal(xb, p.Xa))
    except Exception as e:
        print(m, "EXC", type(e).__name__, str(e)[:200])
