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
