#!/usr/bin/env python3
"""The one-launch analysis (oisat_batch_analyse, OISAT_DAG_SOLVE=1) against the default (task-graph factorization, then the
lock-step solve) on random localised months: grids, observation counts and layouts, correlation lengths, tile sizes,
refinement depths, field types -- the two must give the same BITS (tests/test_gpu_round4.py pins four cases).
usage (GPU box): python tools/fuzz_one_launch.py [cases] [seed]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oi-sat-gmi_amd")]
import numpy as np                                              # noqa: E402
from oisatgmi import _hip, dense, synthetic as syn             # noqa: E402


def fields(p, L, refine, one_launch, dtype, tile_deg, streams):
    os.environ["OISAT_DAG_SOLVE"] = "1" if one_launch else "0"
    ta = dense.TiledAnalysis(p.lat, p.lon, tile_deg=tile_deg, halo_km=3 * L, dtype=dtype, streams=streams)
    ta.load(p.Xa, p.Sa, p.obs_lat, p.obs_lon, p.obs_y, p.obs_var)
    used = bool(ta.factor.one_launch) if ta.factor is not None else False
    status = "ok"
    try:
        ta.run(L, refine=refine, check_pd=True)
        ta.run(L, refine=refine)
    except _hip.OisatError as e:
        status = str(e)[:70]
    xa, inc = ta.download()
    zs = [pl.download_z() for pl in ta.plans if pl is not None]
    ta.close()
    return xa, inc, zs, used, status


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    bad, ran = [], 0
    for c in range(n):
        ny, nx = [(36, 72), (72, 144), (90, 180), (120, 240)][int(rng.integers(0, 4))]
        m = int(rng.choice([300, 1500, 4000, 9000, 16000]))
        L = float(rng.choice([150.0, 250.0, 340.0, 500.0, 800.0]))
        refine = int(rng.integers(0, 4))
        dtype = np.float64 if rng.integers(0, 3) == 0 else np.float32
        tile_deg = float(rng.choice([20.0, 30.0, 45.0, 60.0]))
        streams = int(rng.choice([1, 4, 12]))
        species = str(rng.choice(["NO2", "HCHO", "O3"]))
        seed = int(rng.integers(1, 10 ** 6))
        p = syn.point_obs_case(ny, nx, m, seed, swaths=bool(rng.integers(0, 2)), species=species)
        prm = dict(grid=(ny, nx), m=m, L=L, refine=refine, dtype=np.dtype(dtype).name, tile_deg=tile_deg, streams=streams, species=species, seed=seed)
        a = fields(p, L, refine, True, dtype, tile_deg, streams)
        b = fields(p, L, refine, False, dtype, tile_deg, streams)
        same = a[4] == b[4] and np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True) and len(a[2]) == len(b[2]) and \
            all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a[2], b[2]))
        ran += int(a[3])
        if not same:
            bad.append((prm, a[4], b[4]))
        print(f"case {c}: {prm} one launch used: {a[3]}, status {a[4]!r}; same bits: {same}", flush=True)
    print(f"{n} cases, {ran} of them through the one-launch path; mismatches: {len(bad)}")
    for x in bad:
        print("MISMATCH", x)
    return min(len(bad), 255)


if __name__ == "__main__":
    sys.exit(main())
