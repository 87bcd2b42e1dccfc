"""profiling aid: field error of the dense analysis against the float64 oracle as a function of the refinement tolerance"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oracle import oi_oracle as orc
from oisatgmi import synthetic as syn, dense
c = syn.diag_case(360, 720, 10000, 2001)
lat, lon = syn.global_grid(360, 720)
Y = np.where(c.Y < 0, 0.0, c.Y)
obs = np.isfinite(Y) & np.isfinite(c.So)
cell = np.flatnonzero(obs.ravel())
s = 1.6
rng = np.random.default_rng(11)
others = rng.choice(np.flatnonzero(~obs.ravel()), 3000, replace=False)
sub = np.concatenate([cell, others])
ref = orc.dense_oi(lat.ravel()[sub], lon.ravel()[sub], c.Xa.ravel()[sub], (0.5 * c.Xa.ravel()[sub]) ** 2, lat.ravel()[cell], lon.ravel()[cell],
                   np.arange(cell.size), Y.ravel()[cell], c.So.ravel()[cell], 500.0, scale=s)
fs = np.abs(c.Xa).max()
for dt in (np.float64, np.float32):
    for tol in (0.0, 1e-8, 1e-7, 1e-6):
        xb, inc, info = dense.OI_dense(c.Xa.astype(dt), Y.copy(), ((0.5 * c.Xa) ** 2).astype(dt), c.So.copy(), lat, lon, 500.0, scale=s, refine=3, dtype=dt, tol=tol)
        e = np.abs(xb.ravel()[sub] - ref["xa"]).max() / fs
        print(np.dtype(dt).name, "tol", tol, "resid", ["%.1e" % r for r in info["residuals"]], "max field err / scale %.2e" % e)
