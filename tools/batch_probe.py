"""profiling aid: the lock-step factorization of the 48 tiles (or the 2 caps) of a config-3 month ALONE, per launch shape:
K, tiles, time, TFLOP/s.   usage: OISAT_PROF_DETAIL=1 python tools/batch_probe.py tiles|caps"""
import os, sys, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
which = sys.argv[1] if len(sys.argv) > 1 else "tiles"
ctx = _hip.context(); ctx.own_stream()
p = syn.point_obs_case(720, 1440, 100000, 4000, swaths=True)
ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=12)
ta.prepare(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
sizes = {ti: int(ta.tiles[ti]["obs"].size) for ti in ta.live}
big = sorted(sizes.values())[-2]
only = [ti for ti in ta.live if (sizes[ti] >= big) == (which == "caps")]
ta.prepare(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var, only=only)
ta.build()
ta.run(300.0, refine=2, check_pd=True)
g = ta.factor.ctxs[0]
import time
for rep in range(2):
    ta.pool.enqueue(ta._per_lane(lambda q: q.run_build(300.0))); ta.pool.sync()
    g.prof_reset(); g.prof_enable(True)
    t0 = time.perf_counter()
    g.check(g.lib.oisat_batch_potrf(g.h, ta.factor.ids[0], None)); g.sync()
    el = time.perf_counter() - t0
    prof = g.prof_collect(); g.prof_enable(False)
flops = sum(s ** 3 / 3.0 for ti, s in sizes.items() if ti in only)
print("%s: %d systems, %.2f TFLOP, %.2f ms wall (profiled) = %.1f TFLOP/s" % (which, len(only), flops / 1e12, 1e3 * el, flops / el / 1e12))
rows = []
for k, v in prof.items():
    m = re.match(r"(\w+) K(\d+) t(\d+) n(\d+)", k)
    if m:
        K, t = int(m.group(2)), int(m.group(3))
        fl = 2.0 * 128 * 128 * K * t * v["launches"]
        rows.append((v["total_ms"], k, v["launches"], fl / (v["total_ms"] * 1e-3) / 1e12))
    else:
        rows.append((v["total_ms"], k, v["launches"], 0.0))
tot = sum(r[0] for r in rows)
print("sum of kernel times %.2f ms" % tot)
for ms, k, n, tf in sorted(rows, reverse=True)[:40]:
    print("%8.3f ms  x%-4d %6.1f TFLOP/s  %s" % (ms, n, tf, k))
