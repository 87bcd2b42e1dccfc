import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
ctx = _hip.context()
p = syn.point_obs_case(360, 720, 10000, 4000)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=10000, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa); plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
for _ in range(4):
    plan.run(500.0, refine=1)
ctx.sync()
