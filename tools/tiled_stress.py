"""stress: a localised (tiled, batched) month repeated next to a dense analysis on another stream and bursts of memory traffic;
fields compared bit for bit with the first run.  usage: python tools/tiled_stress.py [seconds]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
ctx = _hip.context()
dev = ctx.device
stop = False
mismatch, counts = [], {}


def dense_worker():
    c = _hip.Context(dev).own_stream()
    c.bind_thread()
    p = syn.point_obs_case(180, 360, 3000, 31)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=3000, dtype=np.float32, ctx=c)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
    plan.run(500.0, refine=2, check_pd=True)
    ref = plan.download()[0].view(np.uint32).copy()
    n = 0
    while not stop:
        for _ in range(3):
            plan.run(500.0, refine=2)
        if not np.array_equal(plan.download()[0].view(np.uint32), ref):
            mismatch.append(("dense", n))
        n += 3
    counts["dense"] = n


def noise():
    c = _hip.Context(dev).own_stream()
    c.bind_thread()
    b = c.alloc(256 << 20)
    n = 0
    while not stop:
        for _ in range(20):
            c.check(c.lib.oisat_memset(c.h, b.ptr, n & 255, b.nbytes))
        c.sync()
        time.sleep(0.003 * (n % 3))
        n += 1


threads = [threading.Thread(target=dense_worker), threading.Thread(target=noise)]
for t in threads:
    t.start()
p = syn.point_obs_case(180, 360, 20000, 4000, swaths=True)
ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=6)
ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
ta.run(300.0, refine=2, check_pd=True)
ref = [a.view(np.uint32).copy() for a in ta.download()]
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    ta.run(300.0, refine=2)
    got = [a.view(np.uint32) for a in ta.download()]
    if not (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])):
        mismatch.append(("tiled", n, int((got[0] != ref[0]).sum())))
    n += 1
counts["tiled months"] = n
stop = True
for t in threads:
    t.join()
ta.close()
print("runs compared bitwise with the first:", counts, "tiles:", len(ta.tiles))
print("MISMATCHES:" if mismatch else "no mismatch", mismatch[:10])
sys.exit(1 if mismatch else 0)
