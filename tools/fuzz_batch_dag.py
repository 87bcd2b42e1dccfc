#!/usr/bin/env python3
"""Random batches through the task-graph Cholesky (oisat_batch_potrf as ONE launch: chains, pairs of big systems, waves of small
ones, the CU reservation) against every member's own factorization by the recursion -- the size mixes between the two that
tests/test_gpu_dag.py pins.  usage (GPU box): python tools/fuzz_batch_dag.py [batches] [seed]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np                                              # noqa: E402
from oisatgmi import _hip, dense, synthetic as syn             # noqa: E402


def system(ctx, m, seed, L_km):
    p = syn.point_obs_case(36, 72, m, seed)
    cell = dense.regular_grid_cell(p.lat, p.lon, p.obs_lat, p.obs_lon)
    keep = (ctx.upload(dense.unit_vectors(p.obs_lat, p.obs_lon)), ctx.upload(np.sqrt(p.Sa.ravel())[cell], dtype=np.float64),
            ctx.upload(p.obs_var, dtype=np.float64))
    mp = -(-m // 128) * 128
    return keep, mp


def build(ctx, keep, m, mp, L_km, S):
    ctx.check(ctx.lib.oisat_cov_build(ctx.h, keep[0].ptr, keep[1].ptr, keep[2].ptr, m, dense.decay_constant(L_km), S.ptr, mp))


def draw_sizes(rng):
    kind = rng.choice(["small", "big_pair", "equal", "tiny", "one_big", "many"])
    if kind == "small":
        return [int(x) for x in rng.integers(60, 1300, size=rng.integers(2, 20))]
    if kind == "big_pair":
        return [int(x) for x in rng.integers(2600, 4200, size=rng.integers(2, 5))] + [int(x) for x in rng.integers(100, 900, size=rng.integers(0, 18))]
    if kind == "equal":
        return [int(rng.integers(129, 1500))] * int(rng.integers(2, 24))
    if kind == "tiny":
        return [int(x) for x in rng.integers(1, 129, size=rng.integers(2, 40))]
    if kind == "one_big":
        return [int(rng.integers(3000, 6000))] + [int(x) for x in rng.integers(64, 700, size=rng.integers(1, 30))]
    return [int(x) for x in rng.integers(100, 520, size=rng.integers(30, 70))]


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = _hip.context()
    lib = ctx.lib
    bad, members = [], 0
    for b in range(nb):
        sizes = draw_sizes(rng)
        L_km = float(rng.choice([200.0, 500.0, 900.0]))
        mats, refs = [], []
        for k, m in enumerate(sizes):
            keep, mp = system(ctx, m, 10000 * b + k, L_km)
            S1, S2 = ctx.alloc(mp * mp * 4), ctx.alloc(mp * mp * 4)
            ctx.check(lib.oisat_set_task_graph(ctx.h, 0))
            build(ctx, keep, m, mp, L_km, S1)
            info = C.c_int(-1)
            ctx.check(lib.oisat_potrf(ctx.h, S1.ptr, m, mp, C.byref(info)))
            refs.append((np.tril(ctx.download(S1.ptr, (mp, mp), np.float32)), info.value))
            S1.free()
            build(ctx, keep, m, mp, L_km, S2)
            mats.append((S2, ctx.alloc(mp * 128 * 4), m, mp, keep))
        n = len(mats)
        Sp = (C.c_void_p * n)(*[a[0].ptr for a in mats])
        Tp = (C.c_void_p * n)(*[a[1].ptr for a in mats])
        mm = (C.c_int64 * n)(*[a[2] for a in mats])
        ld = (C.c_int64 * n)(*[a[3] for a in mats])
        bid = C.c_int(-1)
        ctx.check(lib.oisat_set_task_graph(ctx.h, 1))
        ctx.check(lib.oisat_batch_create(ctx.h, n, Sp, mm, ld, Tp, C.byref(bid)))
        graph = C.c_int(0)
        ctx.check(lib.oisat_batch_is_task_graph(ctx.h, bid.value, C.byref(graph)))
        first = None
        for rep in range(2):
            if rep:
                for (S2, T, m, mp, keep) in mats:
                    build(ctx, keep, m, mp, L_km, S2)
            info2 = (C.c_int * 2)(-1, -1)
            ctx.check(lib.oisat_batch_potrf(ctx.h, bid.value, info2))
            got = [np.tril(ctx.download(a[0].ptr, (a[3], a[3]), np.float32)) for a in mats]
            if any(r[1] != 0 for r in refs):                 # a member the recursion calls not positive definite: both must say so
                if info2[0] == 0:
                    bad.append((b, sizes, "batch reports no bad pivot, the recursion does"))
                break
            if info2[0] != 0:
                bad.append((b, sizes, f"batch info {list(info2)}"))
                break
            for k, (g, (ref, _)) in enumerate(zip(got, refs)):
                err = np.abs(g - ref).max() / np.abs(ref).max()
                if not np.isfinite(g).all() or err > 2e-5:
                    bad.append((b, sizes, f"member {k} (m = {sizes[k]}): {err:.2e}"))
            if rep == 0:
                first = got
            elif not all(np.array_equal(x, y) for x, y in zip(first, got)):
                bad.append((b, sizes, "second run differs from the first"))
        members += n
        ctx.check(lib.oisat_batch_destroy(ctx.h, bid.value))
        for a in mats:
            a[0].free(); a[1].free()
            for kk in a[4]:
                kk.free()
        st = ctx.solve_status(clear=True)
        if not st.clean:
            bad.append((b, sizes, f"status {st}"))
        print(f"batch {b}: {n} systems, {min(sizes)}..{max(sizes)} observations, task graph {bool(graph.value)}, ok so far: {not bad}", flush=True)
    ctx.check(lib.oisat_set_task_graph(ctx.h, -1))
    print(f"{nb} batches, {members} members; mismatches: {len(bad)}")
    for x in bad[:20]:
        print("MISMATCH", x)
    return min(len(bad), 255)


if __name__ == "__main__":
    sys.exit(main())
