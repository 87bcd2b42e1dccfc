"""profiling aid: config 4 (NM months as one MonthTileBatch) -- per group and launch shape: time and TFLOP/s, groups run
one after the other (each ALONE on the GPU).   usage: OISAT_PROF_DETAIL=1 python tools/c4_probe.py [NM]"""
import os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
NM = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ctx = _hip.context(); ctx.own_stream()
L = 300.0
lat2, lon2 = syn.global_grid(720, 1440)
batch = dense.MonthTileBatch(lat2, lon2, 30.0, 3 * L, np.float32, ctx=ctx, streams=12)
for mth in range(NM):
    p = syn.point_obs_case(720, 1440, 100000, 4000 + mth, swaths=True)
    batch.add_month(mth, p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
batch.build()
batch.run(L, refine=2, check_pd=True)
t0 = time.perf_counter(); batch.run(L, refine=2); print("run() %.1f ms for %d months" % (1e3 * (time.perf_counter() - t0), NM))
def per_lane(fn):
    out = [[] for _ in batch.pool.lanes]
    for key, ti in batch._run_order:
        ta = batch.months[key]
        out[ta._lane_of[ti]].append(lambda p=ta.plans[ti]: fn(p))
    return out
batch.pool.enqueue(per_lane(lambda p: p.run_build(L))); batch.pool.sync()
f = batch.factor
for gi, (g, bid, members) in enumerate(zip(f.ctxs, f.ids, f.groups)):
    g.prof_reset(); g.prof_enable(True)
    t0 = time.perf_counter()
    g.check(g.lib.oisat_batch_potrf(g.h, bid, None)); g.sync()
    el = time.perf_counter() - t0
    prof = g.prof_collect(); g.prof_enable(False)
    flops = sum(p.m ** 3 / 3.0 for p in members)
    print("group %d: %d systems of %d..%d obs, %.2f TFLOP, %.1f ms profiled = %.1f TFLOP/s" % (gi, len(members), members[-1].m, members[0].m, flops / 1e12, 1e3 * el, flops / el / 1e12))
    rows = []
    for k, v in prof.items():
        m = re.match(r"(\w+) K(\d+) t(\d+) n(\d+)", k)
        tf = 2.0 * 128 * 128 * int(m.group(2)) * int(m.group(3)) * v["launches"] / (v["total_ms"] * 1e-3) / 1e12 if m else 0.0
        rows.append((v["total_ms"], k, v["launches"], tf))
    print("   sum of kernel times %.1f ms" % sum(r[0] for r in rows))
    lowk = sum(r[0] for r in rows if re.search(r" K(128|256|384) ", r[1]))
    print("   K <= 384 launches: %.1f ms; pair kernels: %.1f ms; potrf_diag %.1f ms" % (lowk, sum(r[0] for r in rows if r[1].startswith("pair")), sum(r[0] for r in rows if r[1].startswith("potrf"))))
    for ms, k, n, tf in sorted(rows, reverse=True)[:14]:
        print("   %8.3f ms  x%-4d %6.1f TFLOP/s  %s" % (ms, n, tf, k))
