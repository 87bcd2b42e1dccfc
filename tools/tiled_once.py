"""profiling aid: one localised (tiled) config-3 month, run a few times (the last pass is what the timeline tools read)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
ctx = _hip.context()
if os.environ.get('OWN_STREAM', '1') == '1':
    ctx.own_stream()          # lane 0 off the NULL stream (a CU-masked stream synchronises with it)
refine = int(os.environ.get("REFINE", "2"))
p = syn.point_obs_case(720, 1440, 100000, 4000, swaths=True)
ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=int(os.environ.get("LANES", "12")))
ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
for _ in range(int(os.environ.get("REPS", "4"))):
    time.sleep(0.05)
    t0 = time.perf_counter(); ta.run(300.0, refine=refine); print("run %.2f ms" % (1e3 * (time.perf_counter() - t0)))
