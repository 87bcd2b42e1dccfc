import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oi-sat-gmi_amd')
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
ctx = _hip.context()
p = syn.point_obs_case(720, 1440, 100000, 4000, swaths=True)
L = 300.0
for streams in (1, 4, 6, 8, 12, 16):
    ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=streams)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    ta.run(L, refine=1, check_pd=True)
    best = 1e9; enq = 0
    for rep in range(3):
        ctx.sync()
        t0 = time.perf_counter()
        for ti in ta._order:
            ta.plans[ti].run(L, refine=1)
        t1 = time.perf_counter()
        for lane in ta.lanes:
            lane.sync()
        t2 = time.perf_counter()
        if t2 - t0 < best:
            best, enq = t2 - t0, t1 - t0
    print("streams %2d: total %.3f s, host enqueue %.3f s, %.1f TF/s" % (streams, best, enq, ta.flops / best / 1e12), flush=True)
    for l in ta.lanes[1:]:
        l.close()
    del ta
