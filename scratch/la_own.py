import sys, time, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = _hip.Context(0).own_stream()
p = syn.point_obs_case(360, 720, m, 4000)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa); plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
L = 500.0 if m <= 10000 else 300.0
res = plan.run(L, refine=1, check_pd=True, want_resid=True)
for _ in range(2): plan.run(L, refine=1)
ctx.sync()
reps = 10 if m <= 20000 else 3
t0 = time.perf_counter()
for _ in range(reps): plan.run(L, refine=1)
ctx.sync()
print("m", m, os.environ.get("OISAT_POTRF"), "free", os.environ.get("OISAT_AUX_FREE_CUS"), "%.3f ms" % ((time.perf_counter() - t0) / reps * 1e3), res[-1])
