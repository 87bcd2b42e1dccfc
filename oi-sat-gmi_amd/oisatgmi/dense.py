"""Dense Gaussian-B optimal interpolation on the MI355X (north-star extension).

    x_a = x_b + B H^T (H B H^T + R)^-1 (y - H x_b)
    B = s D^1/2 C D^1/2,  D = diag(Sa),  C_ij = exp(-chord_ij^2 R_earth^2 / (2 L^2)),  R = diag(So)
    H = selection of the grid cell that contains each observation

The reference's ``OI`` (optimal_interpolation.py:6-52) is the L -> 0 limit of this with one
observation per cell: there C = I and K_ii = s Sa_i / (s Sa_i + So_i) (:27).  There is NO
reference implementation of the dense form, so its parity is pinned only in that limit; away from
it the check is the float64 restatement in ``oracle/`` ("parity unpinned").

Pipeline (all in HBM, csrc/dense_cov.hip + csrc/dense_chol.hip):
``oisat_innovation`` -> ``oisat_cov_build`` (S, fp32) -> ``oisat_potrf`` (MFMA fp32 Cholesky) ->
``oisat_gain_solve`` (triangular solves + float64-residual refinement) -> ``oisat_apply_increment``
(B H^T z generated on the fly, never stored).
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _hip

EARTH_RADIUS_KM = 6371.0
NB = 128                      # Cholesky block edge (csrc/dense_chol.hip)
# relative float64 residual |d - S z| / |d| at which the gain solve stops refining (oisat_gain_solve): the increment is K r
# away from the exact one and |K| <= 1, so the fields are then within 1e-6 |d| -- a tenth of the 1e-5 bar
REFINE_TOL = float(os.environ.get("OISAT_REFINE_TOL", "1e-6"))


def unit_vectors(lat_deg, lon_deg) -> np.ndarray:
    """(3, n) float64 unit vectors, SoA as the kernels want them."""
    la = np.deg2rad(np.ravel(np.asarray(lat_deg, dtype=np.float64)))
    lo = np.deg2rad(np.ravel(np.asarray(lon_deg, dtype=np.float64)))
    return np.ascontiguousarray(np.stack([np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)]))


def morton_order(lat_deg, lon_deg) -> np.ndarray:
    """int32 permutation that lists points along the Z-order curve of (latitude, longitude), 16 bits each: consecutive
    entries are neighbours in space (what ``oisat_set_obs_blocks`` wants)."""
    def spread(v):                                               # 16 bits -> every other bit of 32
        v = v.astype(np.uint32)
        v = (v | (v << 8)) & np.uint32(0x00FF00FF)
        v = (v | (v << 4)) & np.uint32(0x0F0F0F0F)
        v = (v | (v << 2)) & np.uint32(0x33333333)
        v = (v | (v << 1)) & np.uint32(0x55555555)
        return v
    la = np.clip((np.asarray(lat_deg, dtype=np.float64) + 90.0) / 180.0 * 65535.0, 0, 65535)
    lo = np.clip((np.mod(np.asarray(lon_deg, dtype=np.float64) + 180.0, 360.0)) / 360.0 * 65535.0, 0, 65535)
    code = (spread(la.astype(np.uint32)) << np.uint32(1)) | spread(lo.astype(np.uint32))
    return np.argsort(code, kind="stable").astype(np.int32)


def decay_constant(L_km: float) -> float:
    """g in C = exp(-g * chord^2) for a correlation length L (km)."""
    return 0.5 * (EARTH_RADIUS_KM / float(L_km)) ** 2


def regular_grid_cell(lat2, lon2, obs_lat, obs_lon):
    """Flat index of the grid cell containing each observation, for a regular cell-centred grid."""
    lat2 = np.asarray(lat2)
    lon2 = np.asarray(lon2)
    ny, nx = lat2.shape
    dlat = (lat2[-1, 0] - lat2[0, 0]) / max(ny - 1, 1)
    dlon = (lon2[0, -1] - lon2[0, 0]) / max(nx - 1, 1)
    iy = np.clip(np.rint((np.asarray(obs_lat) - lat2[0, 0]) / dlat).astype(np.int64), 0, ny - 1)
    ix = np.clip(np.rint((np.asarray(obs_lon) - lon2[0, 0]) / dlon).astype(np.int64), 0, nx - 1)
    return iy * nx + ix


class DenseAnalysis:
    """Device-resident dense analysis for one grid: upload the grid once, then ``load_obs`` /
    ``run`` per month.  Buffers are sized at ``max_obs`` so nothing is allocated per analysis."""

    def __init__(self, grid_lat, grid_lon, max_obs: int, dtype=np.float32, ctx=None, shared_S=None, out_ptr=None,
                 diag_chunk_rows: int = 0, batched: bool = False):
        """``shared_S``: a factor buffer (``SharedFactor`` or device buffer) shared with other analyses that run one
        after another on the same handle.  ``out_ptr``: device address of 2 n elements where ``xa | inc`` are to be
        written (e.g. a slice of one slab that is gathered over RCCL); by default the plan owns them."""
        self.ctx = ctx or _hip.context()
        self.dt = np.dtype(dtype)
        self.code = _hip.dtype_code(self.dt)
        self.shape = np.shape(grid_lat)
        self._ny, self._nx = (self.shape if len(self.shape) == 2 else (1, int(np.size(grid_lat))))   # cells as a ny x nx grid
        self.n = int(np.size(grid_lat))
        self.max_obs = int(max_obs)
        self.mp_max = -(-self.max_obs // NB) * NB
        c = self.ctx
        self.gxyz = c.upload(unit_vectors(grid_lat, grid_lon))
        self.glat = c.upload(np.ravel(grid_lat), dtype=np.float64)       # latitude window of apply_increment
        item = self.dt.itemsize
        if out_ptr is None:
            # xb | xa | inc; read by torch's RCCL stream when the fields are gathered (parallel.FieldGather)
            self.fields = c.alloc(3 * self.n * item).shared_with_other_streams()
            self.xb_ptr, self.out_ptr = self.fields.at(0), self.fields.at(self.n * item)
        else:
            self.fields = c.alloc(self.n * item)            # xb; xa | inc live in the caller's slab
            self.xb_ptr, self.out_ptr = self.fields.at(0), int(out_ptr)
        self.gsig = c.alloc(self.n * 8)
        m = self.max_obs
        self.oxyz = c.alloc(3 * m * 8)
        self.osig = c.alloc(m * 8)
        self.ovar = c.alloc(m * 8)
        self.ocell = c.alloc(m * 8)
        self.olat = c.alloc(m * 8)
        self.oy = c.alloc(m * 8)
        self.d = c.alloc(m * 8)
        self.z = c.alloc(m * 8)
        # the factor workspace can be shared by analyses that run one after another on the stream (tiles)
        # (a batched plan's S and inverted diagonal blocks are worked on by the BatchedFactor group streams)
        self.S = shared_S if shared_S is not None else c.alloc(self.mp_max * self.mp_max * 4)
        if shared_S is None and batched:
            self.S.shared_with_other_streams()
        if self.S.nbytes < self.mp_max * self.mp_max * 4:
            raise ValueError("shared_S is too small for max_obs")
        # batched factorization (BatchedFactor): the inverted diagonal blocks live in a buffer of this plan, not in the
        # handle's workspace, because many plans of one handle are factored at the same time
        self.tinv = c.alloc(self.mp_max * NB * 4).shared_with_other_streams() if batched else None
        # ... and so do the work vectors and the convergence state of its gain solve (oisat_batch_solve runs on the group's stream)
        self.work = c.alloc(2 * self.mp_max * 8).shared_with_other_streams() if batched else None
        self.state = None
        if batched:
            self.state = c.alloc(256).shared_with_other_streams()
            c.check(c.lib.oisat_memset(c.h, self.state.ptr, 0, 256))
        # the observations along a space-filling curve (compact blocks of rows for the float64 residual)
        self.perm = c.alloc(self.max_obs * 4)
        if batched:
            self.perm.shared_with_other_streams()
        self.m = 0
        self._direct_innovation = False
        # every internal workspace of the solve is sized here, so that run() never allocates (include/oisat.h)
        c.check(c.lib.oisat_dense_reserve(c.h, self.max_obs, int(diag_chunk_rows)))

    # ---- inputs
    def load_background(self, Xa, Sa, scale=1.0):
        c = self.ctx
        c.upload_into(self.xb_ptr, np.ravel(Xa), dtype=self.dt)
        sig = np.sqrt(float(scale) * np.ravel(np.asarray(Sa, dtype=np.float64)))
        self._gsig_host = sig
        c.upload_into(self.gsig.ptr, sig, dtype=np.float64)

    def load_obs_direct(self, obs_lat, obs_lon, obs_sigma_b, obs_var, innovation):
        """Observations whose background value lives outside this plan's grid (tile halos): the caller
        supplies sigma_b at the observation and the innovation d = y - H x_b itself."""
        m = int(np.size(innovation))
        if m < 1 or m > self.max_obs:
            raise ValueError(f"{m} observations; this plan was sized for 1..{self.max_obs}")
        c = self.ctx
        self.m = m
        self.mp = -(-m // NB) * NB
        o = self._sort_by_latitude(obs_lat, obs_lon)
        c.upload_into(self.oxyz.ptr, unit_vectors(np.ravel(obs_lat)[o], np.ravel(obs_lon)[o]))
        c.upload_into(self.osig.ptr, np.ravel(obs_sigma_b)[o], dtype=np.float64)
        c.upload_into(self.ovar.ptr, np.ravel(obs_var)[o], dtype=np.float64)
        c.upload_into(self.d.ptr, np.ravel(innovation)[o], dtype=np.float64)
        self._direct_innovation = True

    def _sort_by_latitude(self, obs_lat, obs_lon):
        """Observations live on the device in ascending-latitude order: the pairs (cell, observation) and
        (observation, observation) whose correlation is above 2^-64 are then contiguous index ranges, which is what the
        latitude windows of ``oisat_apply_increment`` / ``oisat_cov_residual`` skip by.  Per-observation results
        (``download_z``, ``gain_diag``) are handed back in the caller's order."""
        lat = np.ravel(np.asarray(obs_lat, dtype=np.float64))
        self._order = np.argsort(lat, kind="stable")
        self.ctx.upload_into(self.olat.ptr, lat[self._order], dtype=np.float64)
        # ... and the float64 residual takes its blocks of 64 rows along a space-filling curve through them (Morton order of
        # latitude x longitude, as a permutation of the latitude order): neighbours in space, a small bounding sphere
        # (``oisat_set_obs_blocks``)
        lon = np.ravel(np.asarray(obs_lon, dtype=np.float64))[self._order]
        self.ctx.upload_into(self.perm.ptr, morton_order(lat[self._order], lon))
        return self._order

    def _unsort(self, per_obs):
        out = np.empty_like(per_obs)
        out[self._order] = per_obs
        return out

    def load_obs(self, obs_lat, obs_lon, obs_cell, obs_y, obs_var):
        self._direct_innovation = False
        m = int(np.size(obs_y))
        if m < 1 or m > self.max_obs:
            raise ValueError(f"{m} observations; this plan was sized for 1..{self.max_obs}")
        c = self.ctx
        self.m = m
        self.mp = -(-m // NB) * NB
        o = self._sort_by_latitude(obs_lat, obs_lon)
        cell = np.ascontiguousarray(np.ravel(obs_cell)[o], dtype=np.int64)
        c.upload_into(self.oxyz.ptr, unit_vectors(np.ravel(obs_lat)[o], np.ravel(obs_lon)[o]))
        c.upload_into(self.osig.ptr, self._gsig_host[cell], dtype=np.float64)
        c.upload_into(self.ovar.ptr, np.ravel(obs_var)[o], dtype=np.float64)
        c.upload_into(self.ocell.ptr, cell)
        c.upload_into(self.oy.ptr, np.ravel(obs_y)[o], dtype=np.float64)

    # ---- the hot path: everything below runs on the device, enqueued on the handle's stream
    def run(self, L_km: float, refine: int = 2, check_pd: bool = False, want_resid: bool = False, tol=None):
        """``refine``: the most refinement rounds the gain solve may take; it stops earlier once the float64 residual is
        below ``tol`` |d| (default: the handle's, 1e-6 -- ``oisat_set_refine_tol``; 0 runs every round)."""
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.h
        c.check(lib.oisat_set_refine_tol(h, REFINE_TOL if tol is None else float(tol)))      # per run: handles are shared
        m, ld = self.m, self.mp
        g = self._g = decay_constant(L_km)
        item = self.dt.itemsize
        xb, xa, inc = self.xb_ptr, self.out_ptr, self.out_ptr + self.n * item
        if not self._direct_innovation:
            c.check(lib.oisat_innovation(h, self.code, xb, self.ocell.ptr, self.oy.ptr, m, self.d.ptr))
        c.check(lib.oisat_cov_build(h, self.oxyz.ptr, self.osig.ptr, self.ovar.ptr, m, g, self.S.ptr, ld))
        info = C.c_int(0)
        c.check(lib.oisat_potrf(h, self.S.ptr, m, ld, C.byref(info) if check_pd else None))
        resid = (C.c_double * (refine + 1))() if want_resid else None
        c.check(lib.oisat_set_obs_blocks(h, self.perm.ptr, m))                                  # per run: handles are shared
        c.check(lib.oisat_gain_solve(h, self.S.ptr, self.oxyz.ptr, self.osig.ptr, self.ovar.ptr, m, ld, g, self.d.ptr,
                                     int(refine), self.z.ptr, resid, self.olat.ptr))
        c.check(lib.oisat_apply_increment_grid(h, self.code, self.gxyz.ptr, self.gsig.ptr, self._ny, self._nx, self.oxyz.ptr,
                                               self.osig.ptr, self.z.ptr, m, g, xb, xa, inc, self.glat.ptr, self.olat.ptr))
        return list(resid) if want_resid else None

    # ---- the same pipeline in two halves, for lock-step (batched) factorization of many plans: build | factor | solve
    def run_build(self, L_km: float):
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.h
        g = self._g = decay_constant(L_km)
        if not self._direct_innovation:
            c.check(lib.oisat_innovation(h, self.code, self.xb_ptr, self.ocell.ptr, self.oy.ptr, self.m, self.d.ptr))
        c.check(lib.oisat_cov_build(h, self.oxyz.ptr, self.osig.ptr, self.ovar.ptr, self.m, g, self.S.ptr, self.mp))

    def run_solve(self, refine: int = 2, tol=None):
        if self.tinv is None:
            raise ValueError("run_build / run_solve belong to plans created with batched=True (they own their inverted "
                             "diagonal blocks); use run() for a plan that factors on its own handle")
        c, lib, h = self.ctx, self.ctx.lib, self.ctx.h
        m, ld, g = self.m, self.mp, self._g
        item = self.dt.itemsize
        xb, xa, inc = self.xb_ptr, self.out_ptr, self.out_ptr + self.n * item
        c.check(lib.oisat_set_refine_tol(h, REFINE_TOL if tol is None else float(tol)))
        c.check(lib.oisat_factor_adopt(h, self.S.ptr, m, ld, self.tinv.ptr))
        c.check(lib.oisat_set_obs_blocks(h, self.perm.ptr, m))
        c.check(lib.oisat_gain_solve(h, self.S.ptr, self.oxyz.ptr, self.osig.ptr, self.ovar.ptr, m, ld, g, self.d.ptr,
                                     int(refine), self.z.ptr, None, self.olat.ptr))
        c.check(lib.oisat_apply_increment_grid(h, self.code, self.gxyz.ptr, self.gsig.ptr, self._ny, self._nx, self.oxyz.ptr,
                                               self.osig.ptr, self.z.ptr, m, g, xb, xa, inc, self.glat.ptr, self.olat.ptr))

    # ---- posterior diagnostics (after run(); they reuse the factor that run() left in HBM)
    def posterior_error(self, chunk_rows: int = 4096):
        """sqrt(diag(B - B H^T S^-1 H B)) on every grid cell, float32 (n m^2 flop: regional / tiled sizes)."""
        c = self.ctx
        if not hasattr(self, "_err"):
            self._err = c.alloc(self.n * 4)
        g = self._g
        c.check(c.lib.oisat_posterior_error(c.h, self.S.ptr, self.m, self.mp, self.gxyz.ptr, self.gsig.ptr, self.n, 0, self.n,
                                            self.oxyz.ptr, self.osig.ptr, g, int(chunk_rows), self._err.ptr))
        return c.download(self._err.ptr, self.shape, np.float32)

    def gain_diag(self, chunk_rows: int = 4096):
        """diag(K H) at the observations (the averaging kernel of the dense analysis), float64 (m,)."""
        c = self.ctx
        if not hasattr(self, "_ak"):
            self._ak = c.alloc(self.max_obs * 8)
        c.check(c.lib.oisat_gain_diag(c.h, self.S.ptr, self.m, self.mp, self.ovar.ptr, int(chunk_rows), self._ak.ptr))
        return self._unsort(c.download(self._ak.ptr, (self.m,), np.float64))

    # ---- outputs
    def check(self):
        """Wait for this handle's stream and raise ``OisatError`` if any solve enqueued on it since the last check met a
        non-positive pivot or a triangular-solve time-out (``run`` is asynchronous and unchecked by default)."""
        self.ctx.check_solves()

    def download(self):
        self.check()
        out = self.ctx.download(self.out_ptr, (2,) + tuple(self.shape), self.dt)
        return out[0], out[1]

    def download_z(self):
        self.check()
        return self._unsort(self.ctx.download(self.z.ptr, (self.m,), np.float64))

    def download_S(self):
        """S (after ``run``: its Cholesky factor), rows/columns in the device's ascending-latitude order (``self._order``)."""
        return self.ctx.download(self.S.ptr, (self.mp, self.mp), np.float32)

    @staticmethod
    def flops(m: int) -> float:
        """Algorithmic flops of the gain solve as BASELINE.md counts them: m^3/3 + 2 m^2."""
        return m ** 3 / 3.0 + 2.0 * m ** 2


def tile_partition(lat2, lon2, obs_lat, obs_lon, tile_deg=30.0, halo_km=900.0, merge_polar=True):
    """Localised block-B (BASELINE config 3): cut a regular lat/lon grid into tile_deg x tile_deg tiles;
    a tile is analysed with every observation inside the tile or within ``halo_km`` of it (3 L is the
    usual choice: exp(-9/2) = 1 % correlation left).  Returns a list of dicts with the tile's grid
    slices and the indices of its observations.  Longitude wraps; bands whose halo reaches a pole take
    every longitude -- all tiles of such a band then share ONE observation set, i.e. one and the same system
    (H B H^T + R) z = d, so with ``merge_polar`` (default) the band is kept as a single polar-cap tile (all
    longitudes) and that system is built, factored and solved once instead of once per 30 deg of longitude
    (12 times at tile_deg = 30: 87 % of the factorization flops of a 720x1440 / 1e5-observation month)."""
    lat2 = np.asarray(lat2)
    lon2 = np.asarray(lon2)
    ny, nx = lat2.shape
    latc, lonc = lat2[:, 0], lon2[0, :]
    dlat, dlon = abs(latc[1] - latc[0]), abs(lonc[1] - lonc[0])
    ty, tx = max(1, int(round(tile_deg / dlat))), max(1, int(round(tile_deg / dlon)))
    h_lat = np.rad2deg(halo_km / EARTH_RADIUS_KM)
    olat, olon = np.ravel(obs_lat), np.ravel(obs_lon)
    tiles = []
    for y0 in range(0, ny, ty):
        y1 = min(y0 + ty, ny)
        la0, la1 = latc[y0] - dlat / 2, latc[y1 - 1] + dlat / 2
        in_lat = (olat >= la0 - h_lat) & (olat <= la1 + h_lat)
        worst = max(abs(la0 - h_lat), abs(la1 + h_lat))
        polar = worst >= 89.0
        h_lon = 360.0 if polar else h_lat / np.cos(np.deg2rad(worst))
        if polar and merge_polar:
            tiles.append({"rows": (y0, y1), "cols": (0, nx), "obs": np.flatnonzero(in_lat)})
            continue
        for x0 in range(0, nx, tx):
            x1 = min(x0 + tx, nx)
            lo0, lo1 = lonc[x0] - dlon / 2, lonc[x1 - 1] + dlon / 2
            mid, half = 0.5 * (lo0 + lo1), 0.5 * (lo1 - lo0) + h_lon
            dl = np.abs((olon - mid + 180.0) % 360.0 - 180.0)
            sel = np.flatnonzero(in_lat & (dl <= half))
            tiles.append({"rows": (y0, y1), "cols": (x0, x1), "obs": sel})
    return tiles


class BatchedFactor:
    """Lock-step factorization of many plans (``oisat_batch_potrf``): the plans are sorted by size and cut into groups of
    comparable block count (a group's smallest matrix has at least ``ratio`` of the blocks of its largest) -- in a
    720x1440 month: the 48 mid-latitude tiles (31-48 blocks) and the two polar caps (137 blocks).  The groups run side
    by side, one stream each, and each group's solves start when it is factored (``OISAT_BATCH_SCHEDULE``)."""

    def __init__(self, device: int, plans, ratio: float = 0.5):
        groups = []
        order = sorted((p for p in plans if p is not None), key=lambda p: -p.m)
        every = list(order)
        # Task-graph factorization (csrc/dense_dag.inc; OISAT_DAG=0 turns it off): ONE persistent launch factors systems of
        # any mix of sizes -- tiles and polar caps together, the systems entering the launch in waves, every system's chain on
        # a workgroup of its own, the tile tasks drawn from one list -- so a batch of up to 1024 systems is ONE group (larger
        # ones keep the lock-step recursion, and so does a batch whose chains would crowd the handle's CUs: the library
        # decides, oisat_batch_is_task_graph tells).  Measured against the two lock-step groups
        # of the recursion: a month's 50 systems 62 vs 66 ms, a rank's 75 / 150 / 300 units of config 4 0.091 / 0.177 / 0.352 vs
        # 0.103 / 0.185 / 0.357 s, all 600 units 0.700 vs 0.706 s (what the one launch gains its fully exposed solve phase
        # costs).  Several launches side by side are what must be avoided: every launch's chains are resident and its tile
        # tasks starve (seven launches of 96: 1.03 s).
        self.dag = os.environ.get("OISAT_DAG", "-1") != "0" and bool(order) and len(order) <= 1024
        if self.dag:
            groups = [order]
            order = []
        cur = []
        for p in order:
            if cur and p.mp < ratio * cur[0].mp:
                groups.append(cur)
                cur = []
            cur.append(p)
        if cur:
            groups.append(cur)
        # the lock-step recursion (OISAT_DAG=0, or more than 1024 systems) enqueues the group with the longest dependent chain
        # first: in a 720x1440 month the two polar caps (137 diagonal blocks each) are the critical path.  Whatever the order, a
        # group's solves are released when THAT group is factored.  The gain solves and increments of a group run in lock-step on
        # the group's stream right behind its factorization (oisat_batch_solve) when every plan owns its work vectors
        self.batched_solve = all(p.work is not None for p in every)
        # ... or, with the task graph, as tasks of the factorization's own launch (oisat_batch_analyse, OISAT_DAG_SOLVE=1): the
        # same bits in one launch instead of ~15.  Not the default: it ties for one month (61.5-63 vs 62.6 ms) and loses for
        # twelve (0.73 vs 0.71 s) -- the launch is bound by workgroup-slot time, and fp64 VALU work does not overlap with the
        # MFMA K-loop on this chip (csrc/dense_dag.inc "WHAT IT BOUGHT").  The tests run both and compare them bit for bit.
        self.one_launch = self.dag and self.batched_solve and os.environ.get("OISAT_DAG_SOLVE", "0") == "1"
        self.groups = groups
        # schedule (OISAT_BATCH_SCHEDULE): "overlap" (default) -- one stream per group, all groups side by side;
        # "sequential" -- the groups one after the other (each one's dependent chain then runs at its isolated speed and
        # its solves go underneath the next group's factorization).  Same fields either way (tested).
        self.schedule = os.environ.get("OISAT_BATCH_SCHEDULE", "overlap")
        self.ctxs = [_hip.Context(device).own_stream() for _ in self.groups]
        # sharing (oisat_set_share): the group with the longest chain is the critical path of the whole batch -- its waves
        # get priority on the SIMDs they share with the other groups' GEMMs (measured: profiles/EXPERIMENTS.md, round 3)
        if len(self.groups) > 1:
            major = max(range(len(self.groups)), key=lambda gi: self.groups[gi][0].mp)
            for gi, ctx in enumerate(self.ctxs):
                ctx.check(ctx.lib.oisat_set_share(ctx.h, 3 if gi == major else 0, 2))
        self.ids = []
        for g, ctx in zip(self.groups, self.ctxs):
            n = len(g)
            ctx.check(ctx.lib.oisat_set_task_graph(ctx.h, -1 if self.dag else 0))
            Sp = (C.c_void_p * n)(*[p.S.ptr for p in g])
            Tp = (C.c_void_p * n)(*[p.tinv.ptr for p in g])
            mm = (C.c_int64 * n)(*[p.m for p in g])
            ld = (C.c_int64 * n)(*[p.mp for p in g])
            bid = C.c_int(-1)
            ctx.check(ctx.lib.oisat_batch_create(ctx.h, n, Sp, mm, ld, Tp, C.byref(bid)))
            self.ids.append(bid.value)
            yes = C.c_int(0)
            ctx.check(ctx.lib.oisat_batch_is_task_graph(ctx.h, bid.value, C.byref(yes)))
            self.one_launch = self.one_launch and bool(yes.value)      # (the library may have kept the recursion: too many chains for the CUs)
            if self.batched_solve:                          # the solve phase in lock-step too: tell the batch where everything lives
                item = g[0].dt.itemsize
                arr = lambda vals: (C.c_void_p * n)(*vals)      # noqa: E731
                ctx.check(ctx.lib.oisat_batch_set_solve(
                    ctx.h, bid.value, n, arr([p.oxyz.ptr for p in g]), arr([p.osig.ptr for p in g]), arr([p.ovar.ptr for p in g]),
                    arr([p.d.ptr for p in g]), arr([p.olat.ptr for p in g]), arr([p.z.ptr for p in g]),
                    arr([p.work.ptr for p in g]), arr([p.state.ptr for p in g]), arr([p.gxyz.ptr for p in g]),
                    arr([p.gsig.ptr for p in g]), arr([p.glat.ptr for p in g]), (C.c_int64 * n)(*[p.n for p in g]),
                    arr([p.xb_ptr for p in g]), arr([p.out_ptr for p in g]), arr([p.out_ptr + p.n * item for p in g])))
                ctx.check(ctx.lib.oisat_batch_set_grid(ctx.h, bid.value, n, (C.c_int64 * n)(*[p._nx for p in g]),
                                                       arr([p.perm.ptr for p in g])))
        self.group_of = {id(p): gi for gi, g in enumerate(self.groups) for p in g}
        self._threads = None
        # measured, one box (1 month of 720x1440 / 1e5 obs; a rank's eighth of 12 months; all 12 months):
        # overlap 73.6 / 112 / 745 ms, sequential 77.0 / 119 / 746 ms

    def run(self, pool, per_lane_plans, refine, check_pd=False):
        """``per_lane_plans[li]``: the plans of lane li in run order, already BUILT (their S enqueued on the lane).

        Every group's factorization is enqueued at once; the HOST then waits for the groups in completion order
        and enqueues each group's solves on the lanes as it finishes -- it returns when the last group is factored and its
        solves are enqueued.  (A device-side wait would park a barrier packet at the head of every lane's hardware queue for
        the whole factorization, and the command processor polls parked queues at the expense of the running one:
        with 12 parked lanes every kernel of the dependent chain took 40-60 us longer in the rocprofv3 trace --
        potrf_diag 27 -> 64, the panel TRSM 17 -> 80 us -- a third of a polar cap's factorization time.)"""
        sequential = self.schedule == "sequential"

        def enqueue_group(gi):
            g, ctx, bid = self.groups[gi], self.ctxs[gi], self.ids[gi]
            ctx.bind_thread()
            for lane in {id(p.ctx): p.ctx for p in g}.values():
                ctx.wait_for(lane)                          # short: the builds are a few ms
            if sequential and gi > 0:
                ctx.wait_for(self.ctxs[gi - 1])             # one parked queue
            info = (C.c_int * 2)(0, -1)
            if self.one_launch:
                # factorization, gain solves and increments of the whole group as tasks of ONE persistent launch: the solves of
                # the systems factored first run underneath the factorization of the others (csrc/dense_dag.inc)
                ctx.check(ctx.lib.oisat_set_refine_tol(ctx.h, REFINE_TOL))
                ctx.check(ctx.lib.oisat_batch_analyse(ctx.h, bid, g[0].code, g[0]._g, int(refine), info if check_pd else None))
                return
            ctx.check(ctx.lib.oisat_batch_potrf(ctx.h, bid, info if check_pd else None))
            if self.batched_solve:
                ctx.check(ctx.lib.oisat_set_refine_tol(ctx.h, REFINE_TOL))
                ctx.check(ctx.lib.oisat_batch_solve(ctx.h, bid, g[0].code, g[0]._g, int(refine)))

        # One enqueueing host thread per group: a polar-cap factorization is ~700 launches = 20-25 ms of host time inside
        # ONE library call, and a single thread enqueued the groups one after the other -- whichever group came second
        # started that much later (round 2: the caps at 10 ms behind the tiles; caps first: the tiles at 25 ms).
        if len(self.groups) > 1 and not sequential and not check_pd:
            if self._threads is None:
                from concurrent.futures import ThreadPoolExecutor
                self._threads = ThreadPoolExecutor(max_workers=len(self.groups), thread_name_prefix="oisat-group")
            futures = [self._threads.submit(enqueue_group, gi) for gi in range(len(self.groups))]
        else:
            futures = []
            for gi in range(len(self.groups)):
                enqueue_group(gi)
        if self.batched_solve:                              # nothing left for the lanes: wait for the enqueueing threads only
            for f in futures:
                f.exception()
            for f in futures:
                if f.exception() is not None:
                    raise f.exception()
            return
        # release each group's solves as soon as it is factored, in COMPLETION order: poll the group streams
        pending = list(range(len(self.ctxs)))
        while pending:
            done = [gi for gi in pending if (not futures or futures[gi].done()) and not self.ctxs[gi].busy()]
            if not done:
                time.sleep(2e-5)
                continue
            for gi in done:
                if futures and futures[gi].exception() is not None:
                    for f in futures:
                        f.exception()                       # let every enqueueing thread finish before raising
                    raise futures[gi].exception()
                pending.remove(gi)
                pool.enqueue([[(lambda p=p: p.run_solve(refine)) for p in plans if self.group_of[id(p)] == gi]
                              for plans in per_lane_plans])

    def fence(self, pool):
        """Order every lane behind whatever the group streams still have in flight (the lock-step solves of a previous,
        unchecked run read the buffers the next run's builds write).  Device-side, and free when the groups are idle."""
        for lane in pool.lanes:
            for ctx in self.ctxs:
                lane.wait_for(ctx)

    def check(self, what="batched factorization"):
        errors = []
        for ctx in self.ctxs:
            try:
                ctx.check_solves(what)
            except _hip.OisatError as e:
                errors.append(str(e))
        if errors:
            raise _hip.OisatError("; ".join(errors))

    def close(self):
        if self._threads is not None:
            self._threads.shutdown(wait=True)
            self._threads = None
        for ctx, bid in zip(self.ctxs, self.ids):
            if ctx.h is not None:
                ctx.lib.oisat_batch_destroy(ctx.h, bid)
                ctx.close()
        self.ctxs, self.ids = [], []


class SharedFactor:
    """The factor workspace of one lane: analyses that run back to back on a handle share it.  Grow-only; it may be
    re-allocated between (never during) runs, which is why plans hold this object and read ``.ptr`` at run time."""

    def __init__(self, lane):
        self.lane = lane
        self.buf = None
        self.nbytes = 0

    def reserve(self, max_obs: int):
        mp = -(-max(int(max_obs), 1) // NB) * NB
        need = mp * mp * 4
        if need > self.nbytes:
            if self.buf is not None:
                self.lane.sync()
                self.buf.free()
            self.buf = self.lane.alloc(need)
            self.nbytes = need
        return self

    @property
    def ptr(self):
        return self.buf.ptr


class LanePool:
    """Several handles on ONE device, each with its own stream, workspaces and shared factor buffer.  Tiles and
    months are independent, and a 4,000-8,000-observation solve leaves most of the 256 CUs idle on its own."""

    def __init__(self, ctx=None, streams: int = 12):
        self.ctx = ctx or _hip.context()
        self.lanes = [self.ctx] + [_hip.Context(self.ctx.device).own_stream() for _ in range(max(0, int(streams) - 1))]
        self.factors = [SharedFactor(lane) for lane in self.lanes]
        self._threads = None

    @classmethod
    def from_lanes(cls, lanes):
        """A pool over handles the caller already has (each with its own stream); ``close`` leaves them open."""
        self = cls.__new__(cls)
        self.ctx, self.lanes = lanes[0], list(lanes)
        self.factors = [SharedFactor(lane) for lane in self.lanes]
        self._threads = None
        self._borrowed = True
        return self

    def __len__(self):
        return len(self.lanes)

    def assign(self, weights):
        """Longest-processing-time-first: heaviest unit to the least loaded lane.  -> (lane of each unit, run order)."""
        order = sorted(range(len(weights)), key=lambda i: (-float(weights[i]), i))
        load = [0.0] * len(self.lanes)
        lane_of = [0] * len(weights)
        for i in order:
            li = min(range(len(load)), key=load.__getitem__)
            lane_of[i] = li
            load[li] += float(weights[i])
        return lane_of, order

    def sync(self):
        for lane in self.lanes:
            lane.sync()

    def enqueue(self, per_lane):
        """``per_lane[li]``: the callables (each enqueues one unit's kernels) of lane li, in run order.  One host thread
        per lane: a 6,000-observation tile is ~150 kernel launches and a month ~8,000, so a single enqueueing thread --
        ~10 us per launch -- is slower than the GPU (91 ms of launching for 30 ms of work at 720x1440 / 1e5 obs);
        ctypes releases the GIL inside the library calls, and each handle is driven by exactly one thread."""
        work = [(li, fns) for li, fns in enumerate(per_lane) if fns]
        if len(work) <= 1:
            for _, fns in work:
                for fn in fns:
                    fn()
            return
        if self._threads is None:
            from concurrent.futures import ThreadPoolExecutor
            self._threads = ThreadPoolExecutor(max_workers=len(self.lanes), thread_name_prefix="oisat-lane")

        def drive(li, fns):
            self.lanes[li].bind_thread()
            for fn in fns:
                fn()
        futures = [self._threads.submit(drive, li, fns) for li, fns in work]
        errors = [f.exception() for f in futures]        # waits for EVERY lane before anything is raised
        for e in errors:
            if e is not None:
                raise e

    def check(self, what="tiled analysis"):
        """Wait for every lane and read (and clear) every lane's solve status before raising, so that one failed unit
        is reported once and does not poison the next run."""
        errors = []
        for li, lane in enumerate(self.lanes):
            try:
                lane.check_solves(f"{what}, lane {li}")
            except _hip.OisatError as e:
                errors.append(str(e))
        if errors:
            raise _hip.OisatError("; ".join(errors))

    def close(self):
        if self._threads is not None:
            self._threads.shutdown(wait=True)
            self._threads = None
        for f in self.factors:
            if f.buf is not None:
                f.buf.free()
        if getattr(self, "_borrowed", False):
            return
        for lane in self.lanes[1:]:
            lane.close()


def _check_all(what, *checkers):
    """Run every status check (each waits for its streams and clears what it reports), then raise once."""
    errors = []
    for c in checkers:
        if c is None:
            continue
        try:
            c.check(what)
        except _hip.OisatError as e:
            errors.append(str(e))
    if errors:
        raise _hip.OisatError("; ".join(errors))


class TiledAnalysis:
    """Localised block-B analysis: one small dense analysis per tile (its own S = H B H^T + R over the
    tile's observations + halo).  Tiles are independent work units: inside one GPU they are dealt to the lanes of a
    ``LanePool`` (heaviest first, each lane one stream with ONE shared factor workspace); across GPUs
    ``parallel.shard_units`` spreads (month x tile) units -- ``only`` restricts this object to the tiles a rank owns."""

    def __init__(self, grid_lat, grid_lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=None, streams=12, pool=None,
                 merge_polar=True, batched=True):
        """``batched`` (default): every tile keeps its own S and all tiles are factored in lock-step by
        ``oisat_batch_potrf`` (one launch per recursion node for all tiles) between a per-lane build phase and a per-lane
        solve phase; ``batched=False``: each lane runs its tiles' whole pipelines back to back with one shared factor
        buffer (less memory: one S per lane instead of one per tile)."""
        self.merge_polar = bool(merge_polar)
        self.batched = bool(batched)
        self.factor = None
        self._own_pool = pool is None
        self.pool = pool or LanePool(ctx, streams)
        self.ctx = self.pool.ctx
        self.lanes = self.pool.lanes
        self.lat2, self.lon2 = np.asarray(grid_lat), np.asarray(grid_lon)
        self.tile_deg, self.halo_km = float(tile_deg), float(halo_km)
        self.dt = np.dtype(dtype)
        self.plans = []

    def prepare(self, Xa, Sa, obs_lat, obs_lon, obs_y, obs_var, scale=1.0, only=None):
        """Host side: cut the month into tiles, innovation against the GLOBAL background.  ``only``: the tile indices
        this object analyses (default all).  -> the live tile indices and their observation counts."""
        Xa = np.asarray(Xa, dtype=np.float64)
        sig = np.sqrt(float(scale) * np.asarray(Sa, dtype=np.float64))
        olat, olon = np.ravel(obs_lat), np.ravel(obs_lon)
        cell = regular_grid_cell(self.lat2, self.lon2, olat, olon)
        y = np.where(np.ravel(obs_y) < 0, 0.0, np.ravel(obs_y))
        self._host = dict(Xa=Xa, Sa=np.asarray(Sa), scale=float(scale), olat=olat, olon=olon, ovar=np.ravel(obs_var),
                          d=y - Xa.ravel()[cell], s=sig.ravel()[cell])
        self.tiles = tile_partition(self.lat2, self.lon2, olat, olon, self.tile_deg, self.halo_km, self.merge_polar)
        nx = self.lat2.shape[1]
        self._cell, self._Sa, self._scale = cell, np.asarray(Sa), float(scale)
        self._inside = [None] * len(self.tiles)
        for ti, t in enumerate(self.tiles):
            (y0, y1), (x0, x1) = t["rows"], t["cols"]
            c = cell[t["obs"]]
            self._inside[ti] = (c // nx >= y0) & (c // nx < y1) & (c % nx >= x0) & (c % nx < x1)
        keep = set(range(len(self.tiles))) if only is None else set(int(t) for t in only)
        self.live = [ti for ti, t in enumerate(self.tiles) if ti in keep and t["obs"].size]
        self._Xa = Xa
        self.n = int(Xa.size)
        self.flops = sum(DenseAnalysis.flops(int(self.tiles[ti]["obs"].size)) for ti in self.live)
        return self.live, [int(self.tiles[ti]["obs"].size) for ti in self.live]

    def build(self, lane_of=None, order=None, out_ptrs=None, group=True):
        """Device side: one plan per live tile on its lane.  ``lane_of`` / ``order``: per live tile (default: LPT by
        obs^3 over this object's tiles alone); ``out_ptrs``: per live tile, where its ``xa | inc`` go; ``group``: form
        this object's own ``BatchedFactor`` (False when a ``MonthTileBatch`` groups the plans of several months)."""
        h = self._host
        sizes = [int(self.tiles[ti]["obs"].size) for ti in self.live]
        if lane_of is None:
            # batched: the factorization is not a lane's job, what is left per tile grows like m^2 (solves) + n m
            lane_of, order = self.pool.assign([float(m) ** (2 if self.batched else 3) for m in sizes])
        if not self.batched:
            for k, m in enumerate(sizes):
                self.pool.factors[lane_of[k]].reserve(m)
        if self.factor is not None:                      # its group streams may still be working on the OLD plans' factors
            self.factor.close()
            self.factor = None
        self.plans = [None] * len(self.tiles)
        for k, ti in enumerate(self.live):
            t = self.tiles[ti]
            (y0, y1), (x0, x1) = t["rows"], t["cols"]
            li = lane_of[k]
            p = DenseAnalysis(self.lat2[y0:y1, x0:x1], self.lon2[y0:y1, x0:x1], max_obs=sizes[k], dtype=self.dt,
                              ctx=self.lanes[li], shared_S=None if self.batched else self.pool.factors[li],
                              out_ptr=None if out_ptrs is None else out_ptrs[k], batched=self.batched)
            p.load_background(h["Xa"][y0:y1, x0:x1], h["Sa"][y0:y1, x0:x1], scale=h["scale"])
            o = t["obs"]
            p.load_obs_direct(h["olat"][o], h["olon"][o], h["s"][o], h["ovar"][o], h["d"][o])
            self.plans[ti] = p
        self._order = [self.live[k] for k in (order if order is not None else range(len(self.live)))]
        self._lane_of = {ti: lane_of[k] for k, ti in enumerate(self.live)}
        self._host = None
        if self.batched and group and self.live:
            self.factor = BatchedFactor(self.ctx.device, [self.plans[ti] for ti in self.live])

    def load(self, Xa, Sa, obs_lat, obs_lon, obs_y, obs_var, scale=1.0, only=None):
        self.prepare(Xa, Sa, obs_lat, obs_lon, obs_y, obs_var, scale=scale, only=only)
        self.build()

    def _per_lane(self, fn):
        per_lane = [[] for _ in self.lanes]
        for ti in self._order:                           # run order (heaviest first) is kept inside every lane
            per_lane[self._lane_of[ti]].append(lambda p=self.plans[ti]: fn(p))
        return per_lane

    def enqueue(self, L_km, refine=2, check_pd=False):
        if not self._order:                              # not a single observation in any owned tile: x_a = x_b
            return
        if not self.batched:
            self.pool.enqueue(self._per_lane(lambda p: p.run(L_km, refine=refine, check_pd=check_pd)))
            return
        self.factor.fence(self.pool)
        self.pool.enqueue(self._per_lane(lambda p: p.run_build(L_km)))         # innovation, S = H B H^T + R
        plans = [[] for _ in self.lanes]
        for ti in self._order:
            plans[self._lane_of[ti]].append(self.plans[ti])
        self.factor.run(self.pool, plans, refine, check_pd=check_pd)            # lock-step factor | gain solve, increment

    def run(self, L_km, refine=2, check_pd=False):
        """Enqueue every tile on its lane's stream (largest first), wait for all lanes and check their solve status:
        a non-positive pivot or a triangular-solve time-out in ANY tile raises ``OisatError``."""
        self.ctx.sync()                                 # inputs uploaded on the default stream are complete
        self._L, self._refine = float(L_km), int(refine)
        self.enqueue(L_km, refine=refine, check_pd=check_pd)
        _check_all("tiled analysis", self.pool, self.factor)

    def close(self):
        """Release the batch streams / tables and the lane pool (if this object made it)."""
        if self.factor is not None:
            self.factor.close()
            self.factor = None
        self.plans = []
        if self._own_pool:
            self.pool.close()

    def download(self):
        """(xa, inc) on the full grid; cells of tiles this object does not own keep the background / zero."""
        xa = self._Xa.astype(self.dt).copy()
        inc = np.zeros_like(xa)
        for t, p in zip(self.tiles, self.plans):
            if p is None:
                continue
            (y0, y1), (x0, x1) = t["rows"], t["cols"]
            a, b = p.download()
            xa[y0:y1, x0:x1] = a
            inc[y0:y1, x0:x1] = b
        return xa, inc

    def download_error(self, chunk_rows: int = 4096):
        """Posterior error sqrt(diag(B - B H^T S^-1 H B)) on the full grid and diag(K H) per cell (sum over the
        observations inside the cell; 0 where none), tile by tile from each tile's own factor.  A lane shares one
        factor buffer between its tiles, so every tile is re-run (check_pd) right before its diagnostics."""
        err = np.sqrt(self._scale * np.asarray(self._Sa, dtype=np.float64)).astype(np.float32)
        ak = np.zeros(self._Xa.size)
        for ti in self._order:
            t, p = self.tiles[ti], self.plans[ti]
            (y0, y1), (x0, x1) = t["rows"], t["cols"]
            if self.batched:                             # every tile still holds its own factor: adopt, do not redo
                p.ctx.check(p.ctx.lib.oisat_factor_adopt(p.ctx.h, p.S.ptr, p.m, p.mp, p.tinv.ptr))
            else:
                p.run(self._L, refine=self._refine, check_pd=True)
            err[y0:y1, x0:x1] = p.posterior_error(chunk_rows)
            a = p.gain_diag(chunk_rows)
            inside = self._inside[ti]                    # halo observations belong to another tile's cells
            np.add.at(ak, self._cell[t["obs"]][inside], a[inside])
        return err, ak.reshape(self._Xa.shape)


class MonthTileBatch:
    """BASELINE configs[3] -- several monthly analyses cut into (month x tile) work units -- as ONE GPU sees it: the
    units this rank owns, of whatever months, dealt to the lanes of one pool by a single LPT pass and run without any
    host synchronisation in between; every unit writes its ``xa | inc`` tile into one contiguous slab, which is what
    ``parallel`` gathers to rank 0 in a single message.  (Reference: one scheduler job per month,
    run/job_submitter_sbatch.py:45-68; months alone cap 8 GPUs at 12/2 = 6x, hence the finer unit.)"""

    def __init__(self, grid_lat, grid_lon, tile_deg=30.0, halo_km=900.0, dtype=np.float32, ctx=None, streams=12, batched=True):
        self.pool = LanePool(ctx, streams)
        self.batched = bool(batched)
        self.factor = None
        self.ctx = self.pool.ctx
        self.lat2, self.lon2 = np.asarray(grid_lat), np.asarray(grid_lon)
        self.tile_deg, self.halo_km, self.dt = float(tile_deg), float(halo_km), np.dtype(dtype)
        self.months = {}                                # key -> TiledAnalysis restricted to the owned tiles
        self.units = []                                 # (month key, tile index, nobs) in insertion order

    def add_month(self, key, Xa, Sa, obs_lat, obs_lon, obs_y, obs_var, scale=1.0, only=None):
        ta = TiledAnalysis(self.lat2, self.lon2, self.tile_deg, self.halo_km, self.dt, pool=self.pool, batched=self.batched)
        live, sizes = ta.prepare(Xa, Sa, obs_lat, obs_lon, obs_y, obs_var, scale=scale, only=only)
        self.months[key] = ta
        self.units += [(key, ti, m) for ti, m in zip(live, sizes)]

    def build(self, min_slab_elems: int = 0):
        item = self.dt.itemsize
        lane_of, order = self.pool.assign([float(m) ** (2 if self.batched else 3) for (_, _, m) in self.units])
        if not self.batched:
            for li, (_, _, m) in zip(lane_of, self.units):
                self.pool.factors[li].reserve(m)
        self.offsets, total = [], 0                      # element offset of each unit's xa|inc inside the slab
        for key, ti, _ in self.units:
            (y0, y1), (x0, x1) = self.months[key].tiles[ti]["rows"], self.months[key].tiles[ti]["cols"]
            self.offsets.append(total)
            total += 2 * (y1 - y0) * (x1 - x0)
        self.slab_elems = total
        self.slab = self.ctx.alloc(max(total, int(min_slab_elems), 1) * item).shared_with_other_streams()   # lanes write, RCCL reads
        self.ctx.check(self.ctx.lib.oisat_memset(self.ctx.h, self.slab.ptr, 0, self.slab.nbytes))
        k0 = 0
        for key, ta in self.months.items():              # units of one month are contiguous in self.units
            nk = len(ta.live)
            ta.build(lane_of=lane_of[k0:k0 + nk], order=None,
                     out_ptrs=[self.slab.at(self.offsets[k] * item) for k in range(k0, k0 + nk)], group=False)
            k0 += nk
        if self.batched and self.units:                  # one lock-step factorization over the units of ALL months
            self.factor = BatchedFactor(self.ctx.device, [self.months[key].plans[ti] for key, ti, _ in self.units])
        self._run_order = [self.units[i][:2] for i in order]
        self.flops = sum(ta.flops for ta in self.months.values())

    def run(self, L_km, refine=2, check_pd=False, wait=True):
        self.ctx.sync()
        def per_lane(fn):
            out = [[] for _ in self.pool.lanes]
            for key, ti in self._run_order:              # heaviest unit first, across months
                ta = self.months[key]
                out[ta._lane_of[ti]].append(lambda p=ta.plans[ti]: fn(p))
            return out
        if not self._run_order:                          # this rank owns no unit with observations
            pass
        elif self.batched:
            self.factor.fence(self.pool)
            self.pool.enqueue(per_lane(lambda p: p.run_build(L_km)))
            plans = [[] for _ in self.pool.lanes]
            for key, ti in self._run_order:
                ta = self.months[key]
                plans[ta._lane_of[ti]].append(ta.plans[ti])
            self.factor.run(self.pool, plans, refine, check_pd=check_pd)
        else:
            self.pool.enqueue(per_lane(lambda p: p.run(L_km, refine=refine, check_pd=check_pd)))
        if wait:
            self.check()

    def check(self):
        _check_all("month x tile batch", self.pool, self.factor)

    def unit_shape(self, k):
        key, ti, _ = self.units[k]
        (y0, y1), (x0, x1) = self.months[key].tiles[ti]["rows"], self.months[key].tiles[ti]["cols"]
        return (2, y1 - y0, x1 - x0)

    def download_slab(self):
        return self.ctx.download(self.slab.ptr, (self.slab_elems,), self.dt)

    def close(self):
        if self.factor is not None:
            self.factor.close()
            self.factor = None
        for ta in self.months.values():
            ta.plans = []
        self.pool.close()


def OI_dense(Xa, Y, Sa, So, lat, lon, L_km, scale=1.0, refine=2, obs=None, dtype=None, want_error=False, tol=None):
    """Dense-covariance analysis with the reference's gridded argument convention.

    ``Xa, Sa``: (ny, nx) background and its variance; ``Y, So``: (ny, nx) observations and their
    variance, NaN where unobserved (as ``oisatgmi.oi`` passes them, driver.py:110-111);
    ``lat, lon``: (ny, nx) cell centres; ``L_km``: correlation length.  ``obs`` (optional) replaces
    ``Y, So`` by scattered observations: dict(lat, lon, y, var).
    Returns ``(Xb, increment, info)``; unlike the element-wise ``OI`` every cell is analysed
    (unobserved cells get the spread increment instead of NaN).
    """
    Xa = np.asarray(Xa)
    if obs is None:
        Y[Y < 0] = 0.0                                        # same clamp as optimal_interpolation.py:14
        ok = np.isfinite(np.ravel(Y)) & np.isfinite(np.ravel(So)) & np.isfinite(np.ravel(Xa)) & np.isfinite(np.ravel(Sa))
        cell = np.flatnonzero(ok)
        olat, olon = np.ravel(lat)[cell], np.ravel(lon)[cell]
        oy, ovar = np.ravel(Y)[cell], np.ravel(So)[cell]
    else:
        olat, olon = np.ravel(obs["lat"]), np.ravel(obs["lon"])
        oy = np.where(np.ravel(obs["y"]) < 0, 0.0, np.ravel(obs["y"]))
        ovar = np.ravel(obs["var"])
        cell = regular_grid_cell(lat, lon, olat, olon)
    dt = np.dtype(dtype) if dtype is not None else _hip.compute_dtype(Xa)
    if cell.size == 0:                 # a month without a single usable observation: nothing to spread, x_a = x_b
        _hip.context()                 # (still no CPU path: fail here if the library or the GPU is missing)
        extra = {}
        if want_error:
            extra = {"ak": np.full(np.shape(Xa), np.nan), "err": np.sqrt(np.asarray(Sa, dtype=np.float64) * scale),
                     "ak_obs": np.empty(0)}
        return (np.array(Xa, dtype=dt), np.zeros(np.shape(Xa), dtype=dt),
                {"nobs": 0, "residuals": [], "cells": cell, "z": np.empty(0), **extra})
    plan = DenseAnalysis(lat, lon, max_obs=int(cell.size), dtype=dt, diag_chunk_rows=4096 if want_error else 0)
    xa_f = np.where(np.isfinite(Xa), Xa, 0.0)
    sa_f = np.where(np.isfinite(Sa), Sa, 0.0)
    plan.load_background(xa_f, sa_f, scale=scale)
    plan.load_obs(olat, olon, cell, oy, ovar)
    resid = plan.run(L_km, refine=refine, check_pd=True, want_resid=True, tol=tol)
    xb, inc = plan.download()
    extra = {}
    if want_error:                     # the other two members of OI's 4-tuple: averaging kernel and sqrt(Sb)
        # diag(K H) at a cell = sum over the observations INSIDE that cell of diag(H K) at the observation
        # ((B H^T S^-1)[cell(a), a] = (H B H^T S^-1)[a, a]); with one observation per cell -- the reference's
        # case -- this is AK of optimal_interpolation.py:31.  np.add.at: deterministic for repeated cells.
        ak_obs = plan.gain_diag()
        ak_sum = np.zeros(Xa.size)
        np.add.at(ak_sum, cell, ak_obs)
        ak = np.full(Xa.size, np.nan)
        ak[cell] = ak_sum[cell]
        extra = {"ak": ak.reshape(np.shape(Xa)), "err": plan.posterior_error(), "ak_obs": ak_obs}
    bad = ~np.isfinite(Xa)
    if bad.any():
        xb = xb.copy()
        xb[bad] = np.nan
    return xb, inc, {"nobs": int(cell.size), "residuals": resid, "cells": cell, "z": plan.download_z(), **extra}


def OI_tiled(Xa, Y, Sa, So, lat, lon, L_km, tile_deg=30.0, halo_km=None, scale=1.0, refine=2, dtype=None, want_error=False,
             streams=12):
    """Localised block-B analysis with the reference's gridded argument convention (see ``OI_dense``): the grid is cut
    into ``tile_deg`` tiles, each analysed with the observations inside it or within ``halo_km`` (default 3 L).
    Returns ``(Xb, increment, info)``; ``info`` carries ``ak`` / ``err`` when ``want_error``."""
    Xa = np.asarray(Xa)
    Y[Y < 0] = 0.0                                            # same clamp as optimal_interpolation.py:14
    ok = np.isfinite(np.ravel(Y)) & np.isfinite(np.ravel(So)) & np.isfinite(np.ravel(Xa)) & np.isfinite(np.ravel(Sa))
    cell = np.flatnonzero(ok)
    dt = np.dtype(dtype) if dtype is not None else _hip.compute_dtype(Xa)
    if cell.size == 0:
        return OI_dense(Xa, Y, Sa, So, lat, lon, L_km, scale=scale, dtype=dt, want_error=want_error)
    xa_f = np.where(np.isfinite(Xa), Xa, 0.0)
    sa_f = np.where(np.isfinite(Sa), Sa, 0.0)
    ta = TiledAnalysis(lat, lon, tile_deg=tile_deg, halo_km=3.0 * float(L_km) if halo_km is None else halo_km, dtype=dt,
                       streams=streams)
    try:
        ta.load(xa_f, sa_f, np.ravel(lat)[cell], np.ravel(lon)[cell], np.ravel(Y)[cell], np.ravel(So)[cell], scale=scale)
        ta.run(L_km, refine=refine, check_pd=True)
        xb, inc = ta.download()
        extra = {}
        if want_error:
            err, ak = ta.download_error()
            akf = np.full(Xa.size, np.nan)
            akf[cell] = ak.ravel()[cell]
            extra = {"ak": akf.reshape(np.shape(Xa)), "err": err}
    finally:
        ta.close()
    bad = ~np.isfinite(Xa)
    if bad.any():
        xb = xb.copy()
        xb[bad] = np.nan
    return xb, inc, {"nobs": int(cell.size), "cells": cell, "tiles": len(ta.tiles), **extra}
