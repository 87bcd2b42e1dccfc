"""profiling aid (round 4): does VALU work run as a THIRD workgroup per CU beside the two workgroups of a factorization-only
task-graph launch (183 VGPRs / 65.5 KB each; the increment kernel: 96 VGPRs / 26.8 KB)?  Month A's factorization launch on one
stream, month B's 50 increment launches on four others: each alone, then together."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ["OISAT_DAG_SOLVE"] = "0"
import ctypes as C
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
ctx = _hip.context()
ctx.own_stream()
L = 300.0
tas = []
for seed in (4000, 4001):
    p = syn.point_obs_case(720, 1440, 100000, seed, swaths=True)
    ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=ctx, streams=12)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    ta.run(L, refine=2)
    tas.append(ta)
A, B = tas
gA, bidA = A.factor.ctxs[0], A.factor.ids[0]
lanes = [_hip.Context(ctx.device).own_stream() for _ in range(4)]
plansB = [pl for pl in B.plans if pl is not None]


def rebuild_A():
    A.pool.enqueue(A._per_lane(lambda p: p.run_build(L)))
    A.pool.sync()


def factor_A():
    gA.check(gA.lib.oisat_batch_potrf(gA.h, bidA, None))


def inc_B(reps=1):
    for _ in range(reps):
        for k, p in enumerate(plansB):
            c = lanes[k % len(lanes)]
            item = p.dt.itemsize
            c.check(c.lib.oisat_apply_increment_grid(c.h, p.code, p.gxyz.ptr, p.gsig.ptr, p._ny, p._nx, p.oxyz.ptr, p.osig.ptr, p.z.ptr, p.m, p._g,
                                                     p.xb_ptr, p.out_ptr, p.out_ptr + p.n * item, p.glat.ptr, p.olat.ptr))


def sync_all():
    gA.sync()
    for c in lanes:
        c.sync()


for rep in range(2):
    rebuild_A(); sync_all()
    t0 = time.perf_counter(); factor_A(); gA.sync(); fa = time.perf_counter() - t0
    t0 = time.perf_counter(); inc_B(3); sync_all(); ib = (time.perf_counter() - t0) / 3
    rebuild_A(); sync_all()
    t0 = time.perf_counter(); factor_A(); inc_B(3)
    for c in lanes:
        c.sync()
    tb = time.perf_counter() - t0
    gA.sync(); ta_ = time.perf_counter() - t0
    print("factorization alone %.2f ms; a month's increments alone %.2f ms; together: 3 x increments done at %.2f ms, factorization at %.2f ms"
          % (fa * 1e3, ib * 1e3, tb * 1e3, ta_ * 1e3))
