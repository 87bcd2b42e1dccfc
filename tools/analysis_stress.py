"""stress test of the whole dense analysis (build, task-graph factorization, sweeps, float64 residual, increment) under concurrent
uneven load: three analyses of different sizes repeated on three streams next to bursts of memory traffic; every analysis field
is compared bit for bit with the first one of its plan.  usage: python tools/analysis_stress.py [seconds]"""
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


def worker(tag, ny, nx, m, seed, L_km):
    c = _hip.Context(dev).own_stream()
    c.bind_thread()
    p = syn.point_obs_case(ny, nx, m, seed)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=c)
    plan.load_background(p.Xa, p.Sa)
    plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
    plan.run(L_km, refine=2, check_pd=True)
    ref = [a.view(np.uint32).copy() for a in plan.download()]
    zref = plan.download_z().view(np.uint64).copy()
    n = 0
    while not stop:
        for _ in range(3):
            plan.run(L_km, refine=2)
        got = [a.view(np.uint32) for a in plan.download()]
        z = plan.download_z().view(np.uint64)
        if not (np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(z, zref)):
            mismatch.append((tag, n, int((got[0] != ref[0]).sum()), int((z != zref).sum())))
        n += 3
    counts[tag] = n


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


threads = [threading.Thread(target=worker, args=("180x360 m=2500", 180, 360, 2500, 21, 500.0)),
           threading.Thread(target=worker, args=("360x720 m=5000", 360, 720, 5000, 22, 400.0)),
           threading.Thread(target=worker, args=("72x144 m=900", 72, 144, 900, 23, 800.0)),
           threading.Thread(target=noise)]
for t in threads:
    t.start()
time.sleep(budget)
stop = True
for t in threads:
    t.join()
print("analyses compared bitwise with the first:", counts)
print("MISMATCHES:" if mismatch else "no mismatch", mismatch[:10])
sys.exit(1 if mismatch else 0)
