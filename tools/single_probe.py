"""profiling aid: the factorization of ONE system per launch shape (K, tiles, time, TFLOP/s).
usage: OISAT_PROF_DETAIL=1 python tools/single_probe.py M"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = _hip.context()
ny, nx = (360, 720) if m <= 20000 else (720, 1440)
p = syn.point_obs_case(ny, nx, m, 4000, swaths=m > 20000)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
plan = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctx)
plan.load_background(p.Xa, p.Sa)
plan.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
plan.run(500.0, refine=2, check_pd=True)
ctx.prof_reset(); ctx.prof_enable(True)
reps = 5
for _ in range(reps):
    plan.run(500.0, refine=2)
prof = ctx.prof_collect(); ctx.prof_enable(False)
rows = []
for k, v in prof.items():
    mm = re.match(r"(\w+) K(\d+) t(\d+) n(\d+)", k)
    tf = 0.0
    if mm:
        tf = 2.0 * 128 * 128 * int(mm.group(2)) * int(mm.group(3)) * v["launches"] / (v["total_ms"] * 1e-3) / 1e12
    rows.append((v["total_ms"] / reps, k, v["launches"] // reps, tf))
print("m = %d: sum of kernel times %.3f ms per analysis" % (plan.m, sum(r[0] for r in rows)))
for ms, k, n, tf in sorted(rows, reverse=True)[:36]:
    print("%8.3f ms  x%-4d %6.1f TFLOP/s  %s" % (ms, n, tf, k))
