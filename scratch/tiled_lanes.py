import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, dense
import bench
ctx = _hip.context()
p, cell, lat2, lon2 = bench.build_case("config3_720x1440_1e5obs", 4000)
for streams in (1, 2, 4, 6, 8, 12):
    ta = dense.TiledAnalysis(lat2, lon2, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=streams)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    ta.run(300.0, refine=1)
    t0 = time.perf_counter()
    for _ in range(2): ta.run(300.0, refine=1)
    dt = (time.perf_counter() - t0) / 2
    print(f"streams={streams}: {dt*1e3:.1f} ms  {ta.flops/dt/1e12:.1f} TF", flush=True)
    del ta
