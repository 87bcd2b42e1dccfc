"""profiling aid: does a small-LDS kernel (apply_increment of a polar-cap-sized system) run BESIDE a persistent task-graph launch that
holds every workgroup slot, or only after it?  Two handles / streams on one GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
m = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
ctxA = _hip.context()
ctxB = _hip.Context(ctxA.device).own_stream()
ctxA.own_stream()
p = syn.point_obs_case(720, 1440, m, 4000, swaths=True)
cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
A = dense.DenseAnalysis(p.lat, p.lon, max_obs=m, dtype=np.float32, ctx=ctxA)
A.load_background(p.Xa, p.Sa); A.load_obs(p.obs_lat, p.obs_lon, cell, np.where(p.obs_y < 0, 0, p.obs_y), p.obs_var)
q = syn.point_obs_case(360, 720, 10000, 4001)
cellq = dense.regular_grid_cell(q.lat, q.lon, q.obs_lat, q.obs_lon)
B = dense.DenseAnalysis(q.lat, q.lon, max_obs=10000, dtype=np.float32, ctx=ctxB)
B.load_background(q.Xa, q.Sa); B.load_obs(q.obs_lat, q.obs_lon, cellq, np.where(q.obs_y < 0, 0, q.obs_y), q.obs_var)
A.run(300.0, refine=2); B.run(500.0, refine=2); ctxA.sync(); ctxB.sync()
libA, hA, libB, hB = ctxA.lib, ctxA.h, ctxB.lib, ctxB.h
def factorA():
    ctxA.check(libA.oisat_cov_build(hA, A.oxyz.ptr, A.osig.ptr, A.ovar.ptr, A.m, dense.decay_constant(300.0), A.S.ptr, A.mp))
    ctxA.check(libA.oisat_potrf(hA, A.S.ptr, A.m, A.mp, None))
item = 4
def incB(reps=1):
    for _ in range(reps):
        ctxB.check(libB.oisat_apply_increment_grid(hB, B.code, B.gxyz.ptr, B.gsig.ptr, B._ny, B._nx, B.oxyz.ptr, B.osig.ptr, B.z.ptr, B.m,
                                                   dense.decay_constant(500.0), B.xb_ptr, B.out_ptr, B.out_ptr + B.n * item, B.glat.ptr, B.olat.ptr))
rbuf = ctxB.alloc(B.m * 8)
def resB(reps=1):
    for _ in range(reps):
        ctxB.check(libB.oisat_cov_residual(hB, B.oxyz.ptr, B.osig.ptr, B.ovar.ptr, B.m, dense.decay_constant(500.0), B.d.ptr, B.z.ptr, rbuf.ptr, B.olat.ptr))
for name, fn in (("apply_increment x10", lambda: incB(10)), ("cov_residual x10 (10 KB of LDS)", lambda: resB(10))):
    t0 = time.perf_counter(); fn(); ctxB.sync(); alone = time.perf_counter() - t0
    t0 = time.perf_counter(); factorA(); ctxA.sync(); fa = time.perf_counter() - t0
    t0 = time.perf_counter(); factorA(); time.sleep(0.003); fn(); ctxB.sync(); tb = time.perf_counter() - t0; ctxA.sync(); ta = time.perf_counter() - t0
    print("%s: alone %.2f ms; factorization alone %.2f ms; together: B done at %.2f ms (submitted at 3 ms), A done at %.2f ms"
          % (name, alone * 1e3, fa * 1e3, tb * 1e3, ta * 1e3))
