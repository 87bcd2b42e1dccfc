"""One-process-per-GPU sharding of independent analyses over the 8 MI355X of a node.

The reference parallelises this path by submitting one scheduler job per month
(run/job_submitter_sbatch.py:45-68) and one joblib task per input file (reader.py:1405): months,
species and tiles never exchange data.  The MI355X-native equivalent keeps that shape -- there is
NO collective on the data path -- and uses RCCL (torch.distributed backend "nccl") over xGMI only
for what is genuinely shared: one broadcast of the month-invariant lat/lon grid from rank 0, and
the gather of the finished analysis fields to rank 0.  Those messages are a few MB (4.15 MB per
720x1440 fp32 field), i.e. latency-bound, so ring-vs-tree and bucket sizes do not matter here.

Everything below takes/returns torch tensors on whatever device the process group runs on, so the
same code is exercised on CPU with the gloo backend (tests/test_parallel_cpu.py, world_size 2).
"""
from __future__ import annotations

import numpy as np


def shard_units(units, world: int, rank: int, weights=None):
    """Static partition of work units (month x species x tile) over ranks.
    Without weights: round-robin.  With weights (e.g. obs-count^3 of each unit): longest-processing-
    time-first greedy, deterministic, identical on every rank (no communication needed)."""
    units = list(units)
    if weights is None:
        return units[rank::world]
    order = sorted(range(len(units)), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += float(weights[i])
        if r == rank:
            mine.append(i)
    return [units[i] for i in sorted(mine)]


def owner_of(n_units: int, world: int, weights=None):
    """rank that owns each unit under ``shard_units`` (for the gather bookkeeping on rank 0)."""
    owner = [None] * n_units
    for r in range(world):
        for u in shard_units(range(n_units), world, r, weights):
            owner[u] = r
    return owner


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_arrays(arrays, shapes, dtype, device, src=0):
    """Broadcast a list of arrays from ``src`` as ONE message (shapes/dtype known everywhere)."""
    import torch
    dist = _dist()
    sizes = [int(np.prod(s)) for s in shapes]
    buf = torch.empty(sum(sizes), dtype=dtype, device=device)
    if dist.get_rank() == src:
        off = 0
        for a, k in zip(arrays, sizes):
            buf[off:off + k] = torch.as_tensor(np.ascontiguousarray(a).ravel(), dtype=dtype).to(device)
            off += k
    dist.broadcast(buf, src=src)
    out, off = [], 0
    host = buf.cpu().numpy()
    for s, k in zip(shapes, sizes):
        out.append(host[off:off + k].reshape(s).copy())
        off += k
    return out


def broadcast_grid(lat, lon, shape, local_rank=None):
    """The shared model grid: rank 0 -> everyone (RCCL over xGMI when the backend is nccl)."""
    import torch
    dist = _dist()
    dev = torch.device("cuda", local_rank) if (dist.get_backend() == "nccl" and local_rank is not None) else torch.device("cpu")
    lat2, lon2 = broadcast_arrays([lat, lon], [shape, shape], torch.float64, dev)
    return lat2, lon2


def gather_to_root(tensor, dst=0):
    """Every rank contributes one equally-shaped tensor; rank ``dst`` gets the list (rank order)."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "nccl":
        # RCCL: one all_gather into a preallocated slab is the cheapest portable form at MB sizes
        slab = torch.empty((world,) + tuple(tensor.shape), dtype=tensor.dtype, device=tensor.device)
        dist.all_gather_into_tensor(slab, tensor.contiguous())
        return [slab[r] for r in range(world)] if rank == dst else None
    lst = [torch.empty_like(tensor) for _ in range(world)] if rank == dst else None
    dist.gather(tensor, gather_list=lst, dst=dst)
    return lst


class _DevView:
    """Zero-copy torch view of memory owned by the C-ABI library (``__cuda_array_interface__``)."""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class FieldGather:
    """Per-step gather of a DenseAnalysis' (xa, inc) fields to rank 0, straight from the HBM the
    kernels wrote (no staging copy)."""

    def __init__(self, plan, world, rank, local_rank):
        import torch
        self.world, self.rank = world, rank
        item = plan.dt.itemsize
        typestr = "<f4" if item == 4 else "<f8"
        view = _DevView(plan.fields.at(plan.n * item), 2 * plan.n, typestr)
        self.send = torch.as_tensor(view, device=torch.device("cuda", local_rank))
        self.slab = torch.empty((world, 2 * plan.n), dtype=self.send.dtype, device=self.send.device)
        self.shape = plan.shape

    def run(self):
        dist = _dist()
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(self.slab, self.send)       # RCCL, device to device over xGMI
        else:                                                       # rehearsal backends (gloo): list form
            parts = [self.slab[r] for r in range(self.world)]
            dist.all_gather(parts, self.send)
        return self.slab

    def fields(self):
        """rank-major (world, 2, ny, nx) host array of (xa, inc)."""
        return self.slab.cpu().numpy().reshape((self.world, 2) + tuple(self.shape))


def analyse_units(units, analyse, weights=None, result_shape=None, dtype=None, device="cpu"):
    """Run ``analyse(unit) -> tensor(result_shape)`` for this rank's shard and collect every unit's
    result on rank 0 in unit order.  Ranks with fewer units contribute zero slabs on the last rounds
    (a collective needs every rank).  Returns list-of-tensors on rank 0, None elsewhere."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(), dist.get_rank()
    units = list(units)
    mine = shard_units(range(len(units)), world, rank, weights)
    per_rank = [shard_units(range(len(units)), world, r, weights) for r in range(world)]
    rounds = max(len(p) for p in per_rank)
    results = [None] * len(units)
    for k in range(rounds):
        if k < len(mine):
            t = analyse(units[mine[k]])
        else:
            t = torch.zeros(result_shape, dtype=dtype, device=device)
        got = gather_to_root(t, dst=0)
        if rank == 0:
            for r in range(world):
                if k < len(per_rank[r]):
                    results[per_rank[r][k]] = got[r].clone()
    return results if rank == 0 else None
