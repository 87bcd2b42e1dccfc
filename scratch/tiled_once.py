import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
ctx = _hip.context()
p = syn.point_obs_case(720, 1440, 100000, 4000, swaths=True)
ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=int(sys.argv[1]) if len(sys.argv) > 1 else 12)
ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
for _ in range(3):
    t0 = time.perf_counter(); ta.run(300.0, refine=1); print("run %.3f s" % (time.perf_counter() - t0))
