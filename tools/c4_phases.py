"""profiling aid: config 4 (NM months, one GPU) with host syncs BETWEEN the phases: build | lock-step factorizations | solves.
(the product runs them without these syncs: solves of a group start when that group is factored)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import ctypes as C
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
def per_lane(fn):
    out = [[] for _ in batch.pool.lanes]
    for key, ti in batch._run_order:
        ta = batch.months[key]
        out[ta._lane_of[ti]].append(lambda p=ta.plans[ti]: fn(p))
    return out
def sync_all():
    batch.pool.sync()
    [c.sync() for c in batch.factor.ctxs]
f = batch.factor
for rep in range(2):
    sync_all(); t0 = time.perf_counter()
    batch.pool.enqueue(per_lane(lambda p: p.run_build(L))); sync_all(); t1 = time.perf_counter()
    for g, bid in zip(f.ctxs, f.ids):
        g.check(g.lib.oisat_batch_potrf(g.h, bid, None))
    sync_all(); t2 = time.perf_counter()
    if f.batched_solve:                                  # lock-step solves, one group after the other (each alone)
        per_group = []
        for g, bid, members in zip(f.ctxs, f.ids, f.groups):
            prof_on = rep == 1 and os.environ.get("PROF", "0") == "1"
            if prof_on:
                g.prof_reset(); g.prof_enable(True)
            ta = time.perf_counter()
            g.check(g.lib.oisat_batch_solve(g.h, bid, members[0].code, members[0]._g, 2)); g.sync()
            per_group.append(1e3 * (time.perf_counter() - ta))
            if prof_on:
                pr = g.prof_collect(); g.prof_enable(False)
                print("   group of %d:" % len(members), {k: (v["launches"], round(v["total_ms"], 2)) for k, v in pr.items()})
        print("   lock-step solves per group (alone):", ["%.1f ms" % x for x in per_group])
    else:
        batch.pool.enqueue(per_lane(lambda p: p.run_solve(2)))
    sync_all(); t3 = time.perf_counter()
    print("months %d: build %.1f ms | factor (groups side by side) %.1f ms | solves %.1f ms | total %.1f ms" % (NM, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t3 - t0)))
t0 = time.perf_counter(); batch.run(L, refine=2); print("run() %.1f ms" % (1e3 * (time.perf_counter() - t0)))
