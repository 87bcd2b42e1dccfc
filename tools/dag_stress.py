"""stress test of the task graph's hand-over protocol (write-through stores, no acquire on the consumer side): the same systems
factored over and over -- bitwise comparison of every factor with the first one -- while other streams keep the memory system and the
CUs unevenly busy (averaging stacks, a second task-graph factorization of another size, the element-wise OI).  A stale read anywhere
shows as a mismatch.  usage: python tools/dag_stress.py [seconds]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import ctypes as C
import numpy as np
from oisatgmi import _hip, synthetic as syn, dense
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
ctx = _hip.context()
dev = ctx.device


def system(c, m, seed):
    p = syn.point_obs_case(180, 360, m, seed)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    oxyz = c.upload(dense.unit_vectors(p.obs_lat, p.obs_lon))
    osig = c.upload(np.sqrt(p.Sa.ravel())[cell], dtype=np.float64)
    ovar = c.upload(p.obs_var, dtype=np.float64)
    mp = -(-m // 128) * 128
    S = c.alloc(mp * mp * 4)
    def run():
        c.check(c.lib.oisat_cov_build(c.h, oxyz.ptr, osig.ptr, ovar.ptr, m, dense.decay_constant(500.0), S.ptr, mp))
        c.check(c.lib.oisat_potrf(c.h, S.ptr, m, mp, None))
    return run, S, mp, (oxyz, osig, ovar)


stop = False
env_lock = threading.Lock()
mismatch = []
counts = {}


def worker(tag, m, seed, own):
    c = _hip.Context(dev).own_stream() if own else ctx
    c.bind_thread()
    run, S, mp, keep = system(c, m, seed)
    run(); c.sync()
    ref = c.download(S.ptr, (mp, mp), np.float32)
    ref = np.tril(ref).view(np.uint32).copy()
    n = 0
    while not stop:
        for _ in range(5):
            run()
        c.sync()
        got = np.tril(c.download(S.ptr, (mp, mp), np.float32)).view(np.uint32)
        if not np.array_equal(got, ref):
            bad = np.argwhere(got != ref)
            mismatch.append((tag, n, len(bad), tuple(bad[0])))
        n += 5
    counts[tag] = n
    col, nblk, nto = c.solve_status(clear=True)[:3]
    if col or nblk or nto:
        mismatch.append((tag, "status", col, nblk, nto))


def batch_worker(tag, sizes, seed, env):
    """a mixed batch as ONE task-graph launch (waves / chain servers per `env`, read when the plan is made)"""
    c = _hip.Context(dev).own_stream()
    c.bind_thread()
    mats = []
    for k, m in enumerate(sizes):
        p = syn.point_obs_case(180, 360, m, seed + k)
        cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
        oxyz = c.upload(dense.unit_vectors(p.obs_lat, p.obs_lon))
        osig = c.upload(np.sqrt(p.Sa.ravel())[cell], dtype=np.float64)
        ovar = c.upload(p.obs_var, dtype=np.float64)
        mp = -(-m // 128) * 128
        mats.append((c.alloc(mp * mp * 4), c.alloc(mp * 128 * 4), m, mp, oxyz, osig, ovar))
    n = len(mats)
    Sp = (C.c_void_p * n)(*[a[0].ptr for a in mats]); Tp = (C.c_void_p * n)(*[a[1].ptr for a in mats])
    mm = (C.c_int64 * n)(*[a[2] for a in mats]); ld = (C.c_int64 * n)(*[a[3] for a in mats])
    bid = C.c_int(-1)
    c.check(c.lib.oisat_set_task_graph(c.h, 1))
    with env_lock:                                           # (the knobs are read from the environment when the plan is made)
        os.environ.update(env)
        c.check(c.lib.oisat_batch_create(c.h, n, Sp, mm, ld, Tp, C.byref(bid)))
        for k in env:
            del os.environ[k]
    def run():
        for S, T, m, mp, oxyz, osig, ovar in mats:
            c.check(c.lib.oisat_cov_build(c.h, oxyz.ptr, osig.ptr, ovar.ptr, m, dense.decay_constant(500.0), S.ptr, mp))
        c.check(c.lib.oisat_batch_potrf(c.h, bid.value, None))
    def snap():
        return [np.tril(c.download(a[0].ptr, (a[3], a[3]), np.float32)).view(np.uint32).copy() for a in mats]
    run(); c.sync()
    ref = snap()
    it = 0
    while not stop:
        for _ in range(3):
            run()
        c.sync()
        for k, (g, r) in enumerate(zip(snap(), ref)):
            if not np.array_equal(g, r):
                mismatch.append((tag, it, k, int((g != r).sum())))
        it += 3
    counts[tag] = it
    col, nblk, nto = c.solve_status(clear=True)[:3]
    if col or nblk or nto:
        mismatch.append((tag, "status", col, nblk, nto))


def noise():
    c = _hip.Context(dev).own_stream()
    c.bind_thread()
    a = c.upload(np.random.default_rng(0).normal(size=(64 << 20)).astype(np.float32))      # 256 MB
    b = c.alloc(a.nbytes)
    n = 0
    while not stop:
        for _ in range(20):
            c.check(c.lib.oisat_memset(c.h, b.ptr, n & 255, b.nbytes))
        c.sync()
        time.sleep(0.003 * (n % 3))                                                        # bursts: uneven load
        n += 1
    counts["noise"] = n


threads = [threading.Thread(target=worker, args=("A m=2500", 2500, 11, True)),
           threading.Thread(target=worker, args=("B m=4100", 4100, 12, True)),
           threading.Thread(target=worker, args=("C m=900", 900, 13, True)),
           threading.Thread(target=noise)]
if os.environ.get("BATCHES", "1") != "0":
    threads += [threading.Thread(target=batch_worker, args=("batch of eleven", [2100, 1500, 1290, 1000, 777, 640, 300, 257, 129, 128, 100], 900, {})),
                threading.Thread(target=batch_worker, args=("batch of eight", [1800, 900, 800, 700, 600, 500, 400, 300], 950, {}))]
for t in threads:
    t.start()
time.sleep(budget)
stop = True
for t in threads:
    t.join()
print("factorizations compared bitwise with the first:", counts)
print("MISMATCHES:" if mismatch else "no mismatch, no time-out, no bad pivot", mismatch[:10])
sys.exit(1 if mismatch else 0)
