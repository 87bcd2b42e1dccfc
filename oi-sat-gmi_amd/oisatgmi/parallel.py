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
    time-first greedy, deterministic, identical on every rank (no communication needed).  The shard
    comes back in LPT order -- heaviest unit first -- which is also the order to run it in: the big
    factorizations start while every lane is still busy and the small ones fill the tail."""
    units = list(units)
    return [units[i] for i in partition_units(len(units), world, weights)[rank]]


def partition_units(n_units: int, world: int, weights=None):
    """``[unit indices of rank 0, of rank 1, ...]`` -- the whole static partition (every rank can
    compute it: rank 0 needs it to unpack the gather).  Each list is in run order."""
    if weights is None:
        return [list(range(r, n_units, world)) for r in range(world)]
    if len(weights) != n_units:
        raise ValueError(f"{len(weights)} weights for {n_units} units")
    order = sorted(range(n_units), key=lambda i: (-float(weights[i]), i))
    load = [0.0] * world
    parts = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        load[r] += float(weights[i])
        parts[r].append(i)
    return parts


def owner_of(n_units: int, world: int, weights=None):
    """rank that owns each unit under ``shard_units`` (for the gather bookkeeping on rank 0)."""
    owner = [None] * n_units
    for r, part in enumerate(partition_units(n_units, world, weights)):
        for u in part:
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
    """Every rank contributes one equally-shaped tensor; rank ``dst`` -- and only it -- receives them
    (list in rank order, ``None`` elsewhere).  ``dist.gather`` on both backends: over RCCL it is a
    group of point-to-point sends into ``dst``, so the other ranks receive nothing (an all-gather
    would move world x the bytes the analysis needs)."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(), dist.get_rank()
    tensor = tensor.contiguous()
    lst = [torch.empty_like(tensor) for _ in range(world)] if rank == dst else None
    dist.gather(tensor, gather_list=lst, dst=dst)
    return lst


class RemoteRankError(RuntimeError):
    """Another rank failed before the collective; this rank's own work was fine."""


_SLAB_DTYPES = ("float16", "bfloat16", "float32", "float64")          # promotion order of what an analysis returns


def agree(error=None, device="cpu", dtype=None):
    """Every rank enters this (tiny) all-reduce BEFORE the data-path collective, also -- especially -- a rank whose own
    work failed: ``error`` is that rank's exception (or None).  If any rank reports one, every rank raises -- the failing
    rank its own exception, the others ``RemoteRankError`` -- instead of the healthy ranks blocking in the gather until
    the RCCL / gloo time-out.  ``dtype`` (optional torch dtype, None for a rank with nothing to send): the ranks also
    agree on the widest slab dtype, which is returned (None if no rank has one)."""
    import torch
    dist = _dist()
    code = -1
    if dtype is not None:
        name = str(dtype).replace("torch.", "")
        if name in _SLAB_DTYPES:
            code = _SLAB_DTYPES.index(name)
        elif error is None:                                  # an analysis that returned integers, say: this rank's failure -- it
            error = ValueError(f"rank {dist.get_rank()}: an analysis returned dtype {name}; the gathered slab takes one of "
                               f"{', '.join(_SLAB_DTYPES)}")          # still enters the all-reduce, so that every rank raises
    word = torch.tensor([0 if error is None else 1 + dist.get_rank(), code], dtype=torch.int32, device=device)
    dist.all_reduce(word, op=dist.ReduceOp.MAX)
    failed, code = int(word[0]), int(word[1])
    if error is not None:
        raise error
    if failed:
        raise RemoteRankError(f"rank {failed - 1} failed before the gather (see its own traceback); rank {dist.get_rank()} "
                              f"stopped with it instead of waiting for a message that will not come")
    return None if code < 0 else getattr(torch, _SLAB_DTYPES[code])


def checked_gather(send, check, dst=0):
    """``check()`` (e.g. ``MonthTileBatch.check``: waits for the lanes, raises on a failed solve) on every rank, then --
    only if it passed everywhere -- ONE gather of ``send`` to ``dst``."""
    err = None
    try:
        check()
    except Exception as e:                                # noqa: BLE001 -- re-raised by agree() on every rank
        err = e
    if _dist().get_world_size() > 1:
        agree(err, send.device)
    elif err is not None:
        raise err
    return gather_to_root(send, dst=dst)


class DeviceView:
    """Zero-copy torch view of memory owned by the C-ABI library (``__cuda_array_interface__``):
    ``torch.as_tensor(DeviceView(ptr, nelem, "<f4"), device=...)``."""

    def __init__(self, ptr, nelem, typestr):
        self.__cuda_array_interface__ = {"shape": (int(nelem),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


_DevView = DeviceView                                     # round-2 name


class FieldGather:
    """Per-step gather of a DenseAnalysis' (xa, inc) fields to rank 0, straight from the HBM the
    kernels wrote (no staging copy)."""

    def __init__(self, plan, world, rank, local_rank):
        import torch
        self.world, self.rank, self.plan = world, rank, plan
        item = plan.dt.itemsize
        typestr = "<f4" if item == 4 else "<f8"
        view = _DevView(plan.out_ptr, 2 * plan.n, typestr)
        self.send = torch.as_tensor(view, device=torch.device("cuda", local_rank))
        # only the root holds the receive slab
        self.slab = torch.empty((world, 2 * plan.n), dtype=self.send.dtype, device=self.send.device) if rank == 0 else None
        self.shape = plan.shape

    def run(self):
        """xa|inc of this rank's month -> rank 0 (a gather, not an all-gather: nobody else needs them).  The solve
        status of the handle is checked first (one 20-byte read-back), so a failed unchecked run never travels."""
        err = None
        try:
            self.plan.check()
        except Exception as e:                            # noqa: BLE001 -- every rank raises, none waits in the gather
            err = e
        agree(err, self.send.device)
        dist = _dist()
        parts = [self.slab[r] for r in range(self.world)] if self.rank == 0 else None
        dist.gather(self.send, gather_list=parts, dst=0)
        return self.slab if self.rank == 0 else None

    def fields(self):
        """rank-major (world, 2, ny, nx) host array of (xa, inc); rank 0 only."""
        return self.slab.cpu().numpy().reshape((self.world, 2) + tuple(self.shape))


def analyse_units(units, analyse, weights=None, result_shape=None, dtype=None, device="cpu", finish=None):
    """Run ``analyse(unit) -> tensor`` for this rank's shard and collect every unit's result on rank 0
    in unit order.  Returns list-of-tensors on rank 0, None elsewhere.

    There is NO collective inside the unit loop: a rank runs its whole shard, heaviest unit first, at
    its own pace, and the results travel in ONE gather at the end -- so the wall time is the most
    loaded rank's total, not the sum over rounds of the slowest unit of each round (units differ by
    ~90x in cost when they are weighted by obs^3).  ``analyse`` may merely enqueue device work;
    ``finish()`` (optional) is called once after the loop to wait for it before the gather.
    ``result_shape``: one shape for every unit, or a callable ``unit -> shape``."""
    import torch
    dist = _dist()
    world, rank = dist.get_world_size(), dist.get_rank()
    units = list(units)
    parts = partition_units(len(units), world, weights)
    shape_of = result_shape if callable(result_shape) else (lambda u: tuple(result_shape))
    numel = [int(np.prod(shape_of(u))) for u in units]
    cap = max(sum(numel[i] for i in part) for part in parts)          # every rank sends a slab of this size
    # a rank-local failure (an analysis raising, a failed solve surfacing in finish(), a wrong result shape) must not
    # leave the other ranks waiting in the gather: collect it, agree, raise everywhere
    err, got_local = None, []
    try:
        got_local = [analyse(units[i]) for i in parts[rank]]
        if finish is not None:
            finish()
        for i, t in zip(parts[rank], got_local):
            if int(t.numel()) != numel[i]:
                raise ValueError(f"unit {units[i]!r}: analyse returned {tuple(t.shape)}, expected {shape_of(units[i])}")
    except Exception as e:                                # noqa: BLE001
        err = e
    # slab dtype: the caller's, else the widest any rank produced (a rank with an empty shard has none of its own)
    agreed = agree(err, device, dtype if dtype is not None else (got_local[0].dtype if got_local else None))
    dtype = agreed if agreed is not None else torch.float32
    slab = torch.zeros(max(cap, 1), dtype=dtype, device=device)
    off = 0
    for i, t in zip(parts[rank], got_local):
        slab[off:off + numel[i]] = t.reshape(-1).to(dtype)
        off += numel[i]
    got = gather_to_root(slab, dst=0)                                  # the one collective of the data path
    if rank != 0:
        return None
    results = [None] * len(units)
    for r, part in enumerate(parts):
        off = 0
        for i in part:
            results[i] = got[r][off:off + numel[i]].reshape(shape_of(units[i])).clone()
            off += numel[i]
    return results
