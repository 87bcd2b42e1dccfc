"""The N>1 path on CPU: world_size 2 over gloo (the same code runs over RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oisatgmi import parallel


def test_shard_units_partitions_everything():
    units = [(m, t) for m in range(12) for t in range(6)]
    for world in (1, 2, 3, 8):
        seen = []
        for r in range(world):
            seen += parallel.shard_units(units, world, r)
        assert sorted(seen) == sorted(units)
    w = [(i % 7 + 1) ** 3 for i in range(len(units))]
    for world in (2, 8):
        shards = [parallel.shard_units(units, world, r, w) for r in range(world)]
        assert sorted(sum(shards, [])) == sorted(units)
        loads = [sum(w[units.index(u)] for u in s) for s in shards]
        assert max(loads) <= 1.15 * (sum(w) / world) + max(w)          # LPT is within one unit of even
    own = parallel.owner_of(len(units), 8, w)
    assert all(o is not None for o in own)


def test_config4_units_balance_to_more_than_6x_on_8_ranks():
    """BASELINE configs[3]: 12 monthly 720x1440 analyses on 8 GPUs.  Months alone cap the speed-up at 12 / ceil(12/8)
    = 6.0x; with (month x tile) units weighted by obs^3 (the factorization cost) the static LPT partition of
    parallel.shard_units leaves the most loaded rank within a few per cent of the mean: >= 7x by construction
    (no communication on the data path, so the load balance IS the scaling)."""
    from oisatgmi import synthetic as syn, dense
    lat2, lon2 = syn.global_grid(720, 1440)
    units, weights = [], []
    for month in range(12):
        p = syn.point_obs_case(720, 1440, 100000, 4000 + month, swaths=True)
        for ti, t in enumerate(dense.tile_partition(lat2, lon2, p.obs_lat, p.obs_lon, tile_deg=30.0, halo_km=900.0)):
            if t["obs"].size:
                units.append((month, ti))
                weights.append(float(t["obs"].size) ** 3)
    total = sum(weights)
    for world, want in ((2, 1.98), (4, 3.9), (8, 7.0)):
        loads = [sum(weights[units.index(u)] for u in parallel.shard_units(units, world, r, weights)) for r in range(world)]
        assert total / max(loads) >= want, (world, total / max(loads))
    # months as the only unit: the 6.0x ceiling the finer units exist to beat
    mw = [sum(w for (m, _), w in zip(units, weights) if m == month) for month in range(12)]
    loads = [sum(mw[m] for m in parallel.shard_units(range(12), 8, r, mw)) for r in range(8)]
    assert sum(mw) / max(loads) <= 6.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ny, nx = 6, 12
        lat = np.linspace(-80, 80, ny)[:, None] * np.ones((1, nx))
        lon = np.ones((ny, 1)) * np.linspace(-170, 170, nx)[None, :]
        lat2, lon2 = parallel.broadcast_grid(lat if rank == 0 else None, lon if rank == 0 else None, (ny, nx))
        ok_grid = np.array_equal(lat2, lat) and np.array_equal(lon2, lon)

        # 5 months over 2 ranks (uneven): analysis = a deterministic function of (month, grid)
        def analyse(month):
            return torch.as_tensor(np.stack([lat2 * month, lon2 + month]), dtype=torch.float64)

        res = parallel.analyse_units(range(1, 6), analyse, result_shape=(2, ny, nx), dtype=torch.float64)
        ok = ok_grid
        if rank == 0:
            for i, month in enumerate(range(1, 6)):
                ok = ok and np.array_equal(res[i].numpy(), np.stack([lat * month, lon + month]))
        else:
            ok = ok and res is None
        # ragged results (tiles of different sizes) with weights: one gather, unpacked per unit on rank 0
        shapes = {u: (2, u + 1, 3) for u in range(5)}
        rag = parallel.analyse_units(range(5), lambda u: torch.full(shapes[u], float(u)), weights=[5, 1, 4, 2, 3],
                                     result_shape=lambda u: shapes[u], dtype=torch.float64)
        if rank == 0:
            ok = ok and all(tuple(rag[u].shape) == shapes[u] and bool((rag[u] == float(u)).all()) for u in range(5))
        else:
            ok = ok and rag is None
        got = parallel.gather_to_root(torch.full((3,), float(rank)))
        if rank == 0:
            ok = ok and [float(t[0]) for t in got] == [0.0, 1.0]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(2))
    assert got == {0: True, 1: True}


def _sched_worker(rank, world, port, q):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # one heavy unit and eight light ones: LPT gives rank 0 the heavy one and rank 1 the eight light ones
        # (load 8 | 8).  A collective per round would cost max(8,1) + 7 x 1 = 15 units of time; running the
        # shard through and gathering once costs 8.
        weights = [1, 1, 8, 1, 1, 1, 1, 1, 1]
        unit_s = 0.06
        ran = []

        def analyse(u):
            ran.append(u)
            time.sleep(unit_s * weights[u])
            return torch.full((2, 3), float(u))

        dist.barrier()
        t0 = time.perf_counter()
        res = parallel.analyse_units(range(len(weights)), analyse, weights=weights, result_shape=(2, 3), dtype=torch.float32)
        dist.barrier()
        wall = time.perf_counter() - t0
        ok = True
        if rank == 0:
            ok = ok and ran == [2]
            ok = ok and all(float(res[u][0, 0]) == float(u) and tuple(res[u].shape) == (2, 3) for u in range(len(weights)))
        else:
            ok = ok and sorted(ran) == [0, 1, 3, 4, 5, 6, 7, 8] and res is None
        q.put((rank, bool(ok), wall, unit_s))
    finally:
        dist.destroy_process_group()


def test_no_collective_inside_the_unit_loop():
    """the wall time of a sharded run must be the most loaded rank's total (max_r sum_k cost), not
    sum_k max_r cost -- i.e. ranks must not be lock-stepped by a per-round gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sched_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [q.get(timeout=10) for _ in range(2)]
    assert all(g[1] for g in got), got
    unit_s = got[0][3]
    wall = max(g[2] for g in got)
    assert wall < 8 * unit_s * 1.4, (wall, 8 * unit_s)          # lock-stepped rounds would take 15 * unit_s
    assert wall >= 8 * unit_s * 0.98


def _failing_worker(rank, world, port, q):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    try:
        # (1) one rank's analysis raises: EVERY rank leaves analyse_units with an error, quickly
        def analyse(u):
            if u == 3:
                raise FloatingPointError("non-positive pivot in unit 3")
            return torch.full((2, 2), float(u))
        t0 = time.perf_counter()
        try:
            parallel.analyse_units(range(4), analyse, result_shape=(2, 2), dtype=torch.float32)
            out["units"] = "no error"
        except FloatingPointError:
            out["units"] = "own"
        except parallel.RemoteRankError as e:
            out["units"] = "remote:" + str(e)[:6]
        # (2) the failure surfaces in finish() (a failed asynchronous solve) on the other rank
        def finish():
            if rank == 0:
                raise RuntimeError("solve status: time-out")
        try:
            parallel.analyse_units(range(4), lambda u: torch.zeros(3), result_shape=(3,), dtype=torch.float32, finish=finish)
            out["finish"] = "no error"
        except parallel.RemoteRankError:
            out["finish"] = "remote"
        except RuntimeError:
            out["finish"] = "own"
        # (3) checked_gather: check() fails on rank 1 only
        def check():
            if rank == 1:
                raise RuntimeError("lane 3: not positive definite")
        try:
            parallel.checked_gather(torch.zeros(4), check)
            out["gather"] = "no error"
        except parallel.RemoteRankError:
            out["gather"] = "remote"
        except RuntimeError:
            out["gather"] = "own"
        # (3b) ADVICE r3: one rank's analysis returns a dtype the slab cannot take (integers): that rank used to raise BEFORE
        # the all-reduce and leave the other one waiting in it; now it is that rank's failure, carried through the all-reduce
        try:
            parallel.analyse_units(range(2), lambda u: torch.zeros(2, dtype=torch.int32 if rank == 1 else torch.float32), result_shape=(2,))
            out["dtype"] = "no error"
        except parallel.RemoteRankError:
            out["dtype"] = "remote"
        except ValueError:
            out["dtype"] = "own"
        # (4) and the group is still usable afterwards: a clean run, rank 1's shard EMPTY and no dtype given --
        # the slab dtype is agreed across ranks (float64 from rank 0's results), not defaulted per rank
        res = parallel.analyse_units(["only"], lambda u: torch.full((2,), 7.0, dtype=torch.float64), result_shape=(2,))
        out["after"] = (rank != 0 and res is None) or (rank == 0 and res[0].dtype == torch.float64 and float(res[0][1]) == 7.0)
        out["seconds"] = time.perf_counter() - t0
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def test_a_failing_rank_stops_every_rank_instead_of_hanging_the_gather():
    """a rank-local exception right before the only collective used to leave the other ranks blocked in
    dist.gather until the backend's time-out.  Now every rank enters a status all-reduce first and all of them raise."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)                                       # far below any collective time-out
        assert p.exitcode == 0
    got = dict(q.get(timeout=10) for _ in range(2))
    owner3 = parallel.owner_of(4, 2)[3]
    assert got[owner3]["units"] == "own" and got[1 - owner3]["units"].startswith("remote")
    assert got[0]["finish"] == "own" and got[1]["finish"] == "remote"
    assert got[1]["gather"] == "own" and got[0]["gather"] == "remote"
    assert got[1]["dtype"] == "own" and got[0]["dtype"] == "remote"
    assert got[0]["after"] and got[1]["after"]
    assert max(g["seconds"] for g in got.values()) < 20


def _species_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # BASELINE configs[4]: (species x tile) units -- three species' months cut into tiles of different sizes, weighted by
        # obs^3; each unit's result is tagged with its species and tile so that a unit landing in the wrong place shows
        species = ("NO2", "HCHO", "O3")
        units = [(sp, ti) for sp in species for ti in range(7)]
        nobs = {u: 300 + 97 * ((3 * i) % 11) for i, u in enumerate(units)}
        shapes = {u: (2, 3 + u[1] % 3, 4 + (u[1] * 2) % 5) for u in units}
        weights = [float(nobs[u]) ** 3 for u in units]
        ran = []

        def analyse(u):
            ran.append(u)
            sp, ti = u
            return torch.full(shapes[u], 1000.0 * species.index(sp) + ti, dtype=torch.float32)

        res = parallel.analyse_units(units, analyse, weights=weights, result_shape=lambda u: shapes[u], dtype=torch.float32)
        parts = parallel.partition_units(len(units), world, weights)
        ok = sorted(ran) == sorted(units[i] for i in parts[rank]) and [units[i] for i in parts[rank]] == ran     # its LPT shard, heaviest first
        loads = [sum(weights[i] for i in part) for part in parts]
        ok = ok and max(loads) / (sum(loads) / world) < 1.1
        if rank == 0:
            for i, u in enumerate(units):
                sp, ti = u
                ok = ok and tuple(res[i].shape) == shapes[u] and bool((res[i] == 1000.0 * species.index(sp) + ti).all())
            ok = ok and {u[0] for u in ran} != set()      # (which species a rank sees is the partition's business)
        else:
            ok = ok and res is None
        q.put((rank, bool(ok), sorted({u[0] for u in ran})))
    finally:
        dist.destroy_process_group()


def test_species_tile_units_shard_over_two_ranks():
    """BASELINE configs[4] -- the NO2 / HCHO / O3 months as (species x tile) units through the same
    partition / one-gather path as config 4 (run/control_omino2.yml:23, control_omihcho.yml, control_omio3.yml): world
    size 2 over gloo, ragged tiles, every unit back on rank 0 in unit order with its own species' values."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_species_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = [q.get(timeout=10) for _ in range(2)]
    assert all(g[1] for g in got), got
    assert set(got[0][2]) | set(got[1][2]) == {"NO2", "HCHO", "O3"}


def test_shards_come_back_heaviest_first_and_ragged_results_gather():
    w = [3.0, 9.0, 1.0, 27.0, 2.0]
    parts = parallel.partition_units(5, 2, w)
    assert parts == [[3], [1, 0, 4, 2]]
    assert parallel.shard_units("abcde", 2, 1, w) == ["b", "a", "e", "c"]
    assert parallel.partition_units(5, 2) == [[0, 2, 4], [1, 3]]
