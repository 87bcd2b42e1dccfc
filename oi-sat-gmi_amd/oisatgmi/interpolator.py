"""Swath -> model-grid regridding on the MI355X.

Drop-in for ``oisatgmi/interpolator.py`` of the reference (``interpolator``, ``_upscaler``,
``_interpolosis``, ``_boxfilter``, ``_boxfilter2``), all four interpolator types.  What changes
underneath (nearest-neighbour types 2 and 4 first):

* the reference builds a k-d tree over the ~1e6 fine-grid nodes inside EVERY ``_upscaler`` call
  and re-queries the swath tree for every field (interpolator.py:78-88, :28-33).  Here the two
  neighbour searches (swath pixel -> fine node, fine node -> model cell) run once per granule on
  the device (``oisat_nn_query``) and the resulting index vectors are reused by every field;
* all 2-D / per-level fields of a granule are masked, gathered, box-filtered and picked as ONE
  stacked launch each (``oisat_flag_mask`` -> ``oisat_gather_mask`` -> ``oisat_boxfilter_pick``),
  never leaving HBM in between; the box average is evaluated only at the fine nodes the model
  cells pick (interpolator.py:72-91 filters the whole fine grid and throws most of it away).

Type 1 (Delaunay linear, the reference's "recommended" default): the triangulation is built by
qhull on the host exactly as the reference does (``Delaunay(points)``, interpolator.py:153); its
barycentric transforms (``Delaunay.transform``: more host time than qhull itself), point location and
barycentric evaluation -- what ``LinearNDInterpolator`` repeats for every field -- run on the device
for all stacked fields at once (``oisat_tri_transform``, ``oisat_linear_interp``).
Type 3 (``RBFInterpolator(points, Z, neighbors=5)``: thin-plate spline on the 5 nearest pixels): the
reference solves one 8x8 system per distinct neighbourhood in a Python loop, for every field; here one
thread per target finds its neighbours, factors the system once and applies the evaluation weights to
every stacked field (``oisat_rbf_interp``).  Targets the distance mask removes anyway get no value, but
their neighbourhoods are found and factored too (a launch of their own on a coarse point hash), so a
singular one raises ``LinAlgError`` for the whole call exactly where scipy's evaluate-then-mask does.
"""
from __future__ import annotations

import os

import numpy as np

from . import _hip
from .config import satellite_amf, satellite_opt

_I32 = np.dtype(np.int32)
_F64 = np.dtype(np.float64)


def _regrid_dtype() -> np.dtype:
    """The reference promotes every field to float64 on entry (``field*mask`` with a float64 mask,
    interpolator.py:127,:163; float64 box kernels, :40-46), so the device pipeline runs in float64
    unless ``OISAT_DTYPE=f32`` asks for the float32 kernels."""
    return _hip.compute_dtype(np.empty(0, dtype=np.float64))


# --------------------------------------------------------------------------------------------
def _boxfilter(size_kernel_x, size_kernel_y) -> np.ndarray:
    """Mean kernel (interpolator.py:40-42)."""
    return np.full((int(size_kernel_x), int(size_kernel_y)), 1.0) / (size_kernel_x * size_kernel_y)


def _boxfilter2(size_kernel_x, size_kernel_y) -> np.ndarray:
    """Variance-of-the-mean kernel, 1/(n^2) (interpolator.py:44-46)."""
    return np.full((int(size_kernel_x), int(size_kernel_y)), 1.0) / (size_kernel_x * size_kernel_y) ** 2


class NNIndex:
    """Device-resident stand-in for the ``cKDTree`` / ``Delaunay`` objects the reference threads
    through ``_interpolosis``: the source points live in HBM and ``query`` runs the bounded-radius
    exact neighbour search of csrc/regrid.hip.

    Exact ties.  Where a target is equidistant from several points, ``cKDTree.query`` returns whichever of
    them its traversal meets first -- a property of scipy's tree, not of the geometry -- and the reference's
    own settings produce such targets (``grid_size = 1.0`` against a 1.25 / 2.5 degree model longitude spacing,
    reader.py:1209,:1271).  The device search reports them (``oisat_nn_query_ties``) and exactly those targets
    are re-queried against the tree the reference builds over the same point array (interpolator.py:78-88,
    :145-150), once per grid pair; the triangulation of type 1 is taken from qhull in the same way."""

    def __init__(self, lon, lat):
        self.ctx = _hip.context()
        lon = np.ascontiguousarray(np.ravel(lon), dtype=np.float64)
        lat = np.ascontiguousarray(np.ravel(lat), dtype=np.float64)
        self.P = lon.size
        self.lon, self.lat = lon, lat
        self._tree = None
        self._rbf_tree = None
        self.ties_resolved = 0
        self.buf = self.ctx.alloc(2 * self.P * 8)
        self.ctx.upload_into(self.buf.at(0), lon)
        self.ctx.upload_into(self.buf.at(self.P * 8), lat)

    @classmethod
    def from_any(cls, obj):
        if isinstance(obj, cls):
            return obj
        for attr in ("data", "points"):                     # scipy cKDTree / Delaunay
            pts = getattr(obj, attr, None)
            if pts is not None:
                pts = np.asarray(pts)
                return cls(pts[:, 0], pts[:, 1])
        pts = np.asarray(obj)
        return cls(pts[:, 0], pts[:, 1])

    def tree(self):
        """``cKDTree(points)`` over the very array the reference builds it from (interpolator.py:78-82)."""
        if self._tree is None:
            from scipy.spatial import cKDTree
            self._tree = cKDTree(np.column_stack((self.lon, self.lat)))
        return self._tree

    def rbf_tree(self):
        """``KDTree(points)``: the class and defaults ``RBFInterpolator`` builds its own neighbour tree with (not ``cKDTree``:
        the leaf size differs, and with it the pick among equidistant candidates)."""
        if self._rbf_tree is None:
            from scipy.spatial import KDTree
            self._rbf_tree = KDTree(np.column_stack((self.lon, self.lat)))
        return self._rbf_tree

    def query_device(self, tlon, tlat, max_dist, want_dist=False, resolve_ties=True):
        """-> (DeviceBuffer int32[T] of point indices (-1 = beyond max_dist), dist buffer or None)"""
        ctx = self.ctx
        tlon = np.ascontiguousarray(np.ravel(tlon), dtype=np.float64)
        tlat = np.ascontiguousarray(np.ravel(tlat), dtype=np.float64)
        T = tlon.size
        tb = ctx.alloc(2 * T * 8)
        ctx.upload_into(tb.at(0), tlon)
        ctx.upload_into(tb.at(T * 8), tlat)
        idx = ctx.alloc(T * 4)
        dist = ctx.alloc(T * 8) if want_dist else None
        if not resolve_ties:
            ctx.check(ctx.lib.oisat_nn_query(ctx.h, self.buf.at(0), self.buf.at(self.P * 8), self.P, tb.at(0), tb.at(T * 8),
                                             T, float(max_dist), idx.ptr, dist.ptr if dist else None))
            ctx.sync()
            tb.free()
            return idx, dist
        ties = ctx.alloc(T * 4)
        n = _hip.C.c_int64(0)
        ctx.check(ctx.lib.oisat_nn_query_ties(ctx.h, self.buf.at(0), self.buf.at(self.P * 8), self.P, tb.at(0), tb.at(T * 8),
                                              T, float(max_dist), idx.ptr, dist.ptr if dist else None, ties.ptr,
                                              _hip.C.byref(n)))
        ctx.sync()
        tb.free()
        if n.value:
            which = np.sort(ctx.download(ties.ptr, (int(n.value),), _I32))
            _, pick = self.tree().query(np.column_stack((tlon[which], tlat[which])))
            host = ctx.download(idx.ptr, (T,), _I32)
            host[which] = pick.astype(np.int32)
            ctx.upload_into(idx.at(0), host)
            ctx.sync()
            self.ties_resolved += int(n.value)
        ties.free()
        return idx, dist

    def query(self, targets, max_dist=np.inf):
        """``cKDTree.query``-like host result for targets within ``max_dist`` (others: inf, -1)."""
        t = np.asarray(targets, dtype=np.float64)
        idx, dist = self.query_device(t[..., 0], t[..., 1], max_dist, want_dist=True)
        i = self.ctx.download(idx.ptr, t.shape[:-1], _I32)
        d = self.ctx.download(dist.ptr, t.shape[:-1], _F64)
        return d, i


class TriIndex:
    """Device copy of a ``scipy.spatial.Delaunay`` triangulation (simplices, neighbours, barycentric
    transforms, one incident simplex per vertex) for ``oisat_linear_interp``."""

    def __init__(self, tri):
        self.ctx = ctx = _hip.context()
        self.tri = tri
        self.ns = int(tri.simplices.shape[0])
        self.simplices = ctx.upload(tri.simplices, dtype=np.int32)
        self.neighbors = ctx.upload(tri.neighbors, dtype=np.int32)
        self.v2s = ctx.upload(tri.vertex_to_simplex, dtype=np.int32)
        self.P = int(tri.points.shape[0])
        self.ambiguous = 0
        self._barycentric_transforms()
        self.bounds = (_hip.C.c_double * 4)(float(tri.min_bound[0]), float(tri.max_bound[0]), float(tri.min_bound[1]),
                                             float(tri.max_bound[1]))

    def _barycentric_transforms(self):
        """``Delaunay.transform`` without its host cost (three LAPACK calls per simplex under the GIL: 0.55-1.0 s per OMI
        granule, more than qhull): ``oisat_tri_transform`` walks the same 2 x 2 elimination per simplex on the device; the
        simplices it reports as close to scipy's degeneracy limit are handed to scipy's own routine, so which simplices are
        NaN is always scipy's decision.  A triangulation that already carries its transform (a worker computed it) keeps it."""
        ctx, tri = self.ctx, self.tri
        have = getattr(tri, "_transform", self)            # scipy's private cache of the lazily computed property
        if have is self or have is not None:               # another scipy layout, or the transform is there already: take scipy's
            have = tri.transform
            self.transform = ctx.upload(have, dtype=np.float64)
            self.has_degenerate = bool(np.isnan(have[:, 0, 0]).any())
            return
        pts = ctx.upload(tri.points, dtype=np.float64)
        self.transform = ctx.alloc(self.ns * 6 * 8)
        suspects = ctx.alloc(self.ns * 4)
        n = _hip.C.c_int64(0)
        ctx.check(ctx.lib.oisat_tri_transform(ctx.h, pts.ptr, self.P, self.simplices.ptr, self.ns, self.transform.ptr, suspects.ptr,
                                              _hip.C.byref(n)))
        self.has_degenerate = False
        if n.value:
            from scipy.spatial import _qhull
            which = np.sort(ctx.download(suspects.ptr, (int(n.value),), _I32))
            rows = _qhull._get_barycentric_transforms(tri.points, np.ascontiguousarray(tri.simplices[which]), np.finfo(np.float64).eps)
            host = ctx.download(self.transform.ptr, (self.ns, 3, 2), _F64)
            host[which] = rows
            ctx.upload_into(self.transform.at(0), host)
            self.has_degenerate = bool(np.isnan(rows[:, 0, 0]).any())
            tri._transform = host
        pts.free()
        suspects.free()

    def host_transform(self):
        """The triangulation with its transform filled in from the device copy (``find_simplex`` would otherwise compute its
        own on the host, and then not necessarily the very same last bits the device evaluates with)."""
        if getattr(self.tri, "_transform", self) is None:
            self.tri._transform = self.ctx.download(self.transform.ptr, (self.ns, 3, 2), _F64)
        return self.tri

    @classmethod
    def from_points(cls, lon, lat):
        """None when qhull cannot triangulate (the reference returns None for the granule, :151-155)."""
        from scipy.spatial import Delaunay
        pts = np.column_stack((np.ravel(lon), np.ravel(lat))).astype(np.float64)
        try:
            return cls(Delaunay(pts))
        except Exception:
            return None

    def locate(self, tgt_buf, T, nn_idx_buf, tlon, tlat):
        """Simplices for the targets whose location is not unique (on a shared facet / vertex, or found by the brute-force
        scan): scipy evaluates targets one after the other, each walk starting where the previous one ended, so the simplex
        such a target gets -- and, next to a NaN vertex, whether the output is NaN -- is a property of that sequential
        search.  Exactly those targets are therefore located by ``Delaunay.find_simplex`` over the whole target list (same
        walk, same order, same eps as LinearNDInterpolator, interpolator.py:13-15); every other target keeps the device
        walk (a proper triangulation is a partition: strictly inside one simplex means inside no other).  -> DeviceBuffer int32[T] (-2 = walk on the device) or None when no target is ambiguous (any jittered swath)."""
        ctx = self.ctx
        if self.has_degenerate:
            # qhull closed the hull with zero-area simplices (NaN transforms): the triangulation is then not a partition
            # -- large flat simplices overlap their neighbours -- and "the" simplex of a target is whatever the sequential
            # search returns, for any target.  Take all of them from it.
            found = self.host_transform().find_simplex(np.column_stack((np.ravel(tlon), np.ravel(tlat))).astype(np.float64))
            self.ambiguous = int(T)
            return ctx.upload(found.astype(np.int32))
        amb = ctx.alloc(T * 4)
        n = _hip.C.c_int64(0)
        ctx.check(ctx.lib.oisat_linear_locate(ctx.h, tgt_buf.at(0), tgt_buf.at(T * 8), T, nn_idx_buf.ptr, self.v2s.ptr,
                                              self.simplices.ptr, self.neighbors.ptr, self.transform.ptr, self.ns, self.P,
                                              self.bounds, amb.ptr, _hip.C.byref(n)))
        ctx.sync()
        self.ambiguous = int(n.value)
        if not n.value:
            amb.free()
            return None
        which = np.sort(ctx.download(amb.ptr, (int(n.value),), _I32))
        amb.free()
        found = self.host_transform().find_simplex(np.column_stack((np.ravel(tlon), np.ravel(tlat))).astype(np.float64))
        forced = np.full(T, -2, dtype=np.int32)
        forced[which] = found[which]
        return ctx.upload(forced)

    def interpolate(self, dt, values_buf, nfields, tgt_buf, T, nn_idx_buf, forced=None):
        ctx = self.ctx
        out = ctx.alloc(nfields * T * dt.itemsize)
        if forced is not None:
            ctx.check(ctx.lib.oisat_linear_interp_forced(ctx.h, _hip.dtype_code(dt), tgt_buf.at(0), tgt_buf.at(T * 8), T,
                                                         nn_idx_buf.ptr, self.v2s.ptr, self.simplices.ptr, self.neighbors.ptr,
                                                         self.transform.ptr, self.ns, values_buf.ptr, self.P, nfields, out.ptr,
                                                         self.bounds, forced.ptr))
            return out
        ctx.check(ctx.lib.oisat_linear_interp(ctx.h, _hip.dtype_code(dt), tgt_buf.at(0), tgt_buf.at(T * 8), T, nn_idx_buf.ptr,
                                              self.v2s.ptr, self.simplices.ptr, self.neighbors.ptr, self.transform.ptr, self.ns,
                                              values_buf.ptr, self.P, nfields, out.ptr, self.bounds))
        return out


class _RbfTies:
    """Targets whose fifth neighbour is not unique, with the neighbourhoods scipy's own tree gives them: found with the first
    field stack of a (points, targets) pair, reused by the later ones."""

    def __init__(self):
        self.known = False
        self.n = 0
        self.targets = None
        self.ids = None


def _rbf(nn: "NNIndex", dt, values_buf, nfields, tgt_buf, T, idx_buf, cell, memo=None):
    """``RBFInterpolator(points, values, neighbors=5)`` at the targets whose nearest point lies within ``cell``.

    Two things scipy's evaluate-everything-then-mask does are kept although the masked targets get no value here: their
    neighbourhoods are factored too, so that a singular one raises (``oisat_rbf_check_masked``), and where the fifth neighbour
    of an unmasked target is not unique the choice is the one ``RBFInterpolator``'s own ``KDTree(points).query(x, 5)`` makes
    (``oisat_rbf_interp_ties`` reports those targets, the host asks the tree, ``oisat_rbf_interp_forced`` evaluates them).
    Neither depends on the values: ``memo`` (a ``_RbfTies``) carries both over to further field stacks of the same pair."""
    ctx = nn.ctx
    if nn.P < 3:                        # scipy: 3 monomials need 3 points
        raise ValueError("At least 3 data points are required when `degree` is 1 and the number of dimensions is 2.")
    K = int(min(5, nn.P))
    code = _hip.dtype_code(dt)
    px, py, tx, ty = nn.buf.at(0), nn.buf.at(nn.P * 8), tgt_buf.at(0), tgt_buf.at(T * 8)
    nsing = _hip.C.c_int64(0)
    memo = memo if memo is not None else _RbfTies()
    if not memo.known:
        ctx.check(ctx.lib.oisat_rbf_check_masked(ctx.h, px, py, nn.P, tx, ty, T, idx_buf.ptr, float(cell), K, _hip.C.byref(nsing)))
        if nsing.value:
            raise np.linalg.LinAlgError("Singular matrix.")
    out = ctx.alloc(nfields * T * dt.itemsize)
    ties = ctx.alloc(T * 4)
    nties = _hip.C.c_int64(0)
    ctx.check(ctx.lib.oisat_rbf_interp_ties(ctx.h, code, px, py, nn.P, tx, ty, T, idx_buf.ptr, float(cell), K, values_buf.ptr,
                                            nfields, out.ptr, _hip.C.byref(nsing), ties.ptr, _hip.C.byref(nties)))
    if nsing.value:
        raise np.linalg.LinAlgError("Singular matrix.")
    if not memo.known:
        memo.known = True
        memo.n = int(nties.value)
        if memo.n:
            which = np.sort(ctx.download(ties.ptr, (memo.n,), _I32))
            xy = ctx.download(tgt_buf.ptr, (2, T), _F64)[:, which].T
            _, pick = nn.rbf_tree().query(xy, K)
            memo.targets = ctx.upload(which, dtype=_I32)
            memo.ids = ctx.upload(np.ascontiguousarray(pick.reshape(memo.n, K), dtype=np.int32), dtype=_I32)
            nn.ties_resolved += memo.n
    ties.free()
    if memo.n:
        ctx.check(ctx.lib.oisat_rbf_interp_forced(ctx.h, code, px, py, nn.P, tx, ty, T, memo.targets.ptr, memo.ids.ptr, memo.n, K,
                                                  values_buf.ptr, nfields, out.ptr, _hip.C.byref(nsing)))
        if nsing.value:
            raise np.linalg.LinAlgError("Singular matrix.")
    return out


def _gather(ctx, dt, values_buf, P, nfields, idx_buf, T):
    out = ctx.alloc(nfields * T * dt.itemsize)
    ctx.check(ctx.lib.oisat_gather_mask(ctx.h, _hip.dtype_code(dt), values_buf.ptr, P, nfields, idx_buf.ptr, T, out.ptr))
    return out


def _interpolosis(interpol_func, Z: np.ndarray, X: np.ndarray, Y: np.ndarray, interpolator_type: int,
                  dists: np.ndarray, threshold: float) -> np.ndarray:
    """One field, host in / host out (interpolator.py:10-37)."""
    if interpolator_type not in (1, 2, 3, 4):
        raise Exception("other type of interpolation methods has not been implemented yet")
    ctx = _hip.context()
    if interpolator_type == 3:          # interpol_func: the (P, 2) point array (interpolator.py:156-157)
        nn = NNIndex.from_any(interpol_func)
        dt = _hip.compute_dtype(Z)
        Zb = ctx.upload(np.ravel(Z), dtype=dt)
        T = int(np.size(X))
        tb = ctx.alloc(2 * T * 8)
        ctx.upload_into(tb.at(0), np.ravel(X), dtype=np.float64)
        ctx.upload_into(tb.at(T * 8), np.ravel(Y), dtype=np.float64)
        cell = 2.0 * float(threshold)
        idx, _ = nn.query_device(X, Y, cell, resolve_ties=False)     # only decides which targets are skipped
        out = _rbf(nn, dt, Zb, 1, tb, T, idx, cell)
        ZZ = ctx.download(out.ptr, np.shape(X), dt)
        ZZ[np.asarray(dists) > threshold * 2.0] = np.nan
        return ZZ
    if interpolator_type == 1:
        ti = interpol_func if isinstance(interpol_func, TriIndex) else TriIndex(interpol_func)     # a scipy Delaunay
        pts = ti.tri.points
        nn = NNIndex(pts[:, 0], pts[:, 1])
        dt = _hip.compute_dtype(Z)
        Zb = ctx.upload(np.ravel(Z), dtype=dt)
        T = int(np.size(X))
        tb = ctx.alloc(2 * T * 8)
        ctx.upload_into(tb.at(0), np.ravel(X), dtype=np.float64)
        ctx.upload_into(tb.at(T * 8), np.ravel(Y), dtype=np.float64)
        # walk start: nearest pixel of every target (unbounded radius here; the mask comes from `dists`)
        span = float(np.hypot(np.ptp(pts[:, 0]) + np.ptp(np.ravel(X)), np.ptp(pts[:, 1]) + np.ptp(np.ravel(Y)))) + 1.0
        idx, _ = nn.query_device(X, Y, span, resolve_ties=False)     # walk start only
        out = ti.interpolate(dt, Zb, 1, tb, T, idx, ti.locate(tb, T, idx, X, Y))
        ZZ = ctx.download(out.ptr, np.shape(X), dt)
        ZZ[np.asarray(dists) > threshold * 2.0] = np.nan
        return ZZ
    nn = NNIndex.from_any(interpol_func)
    dt = _hip.compute_dtype(Z)
    Zb = ctx.upload(np.ravel(Z), dtype=dt)
    # the mask radius is carried by `dists`; search wide enough to reproduce the unmasked gather
    idx, _ = nn.query_device(X, Y, max(2.0 * float(threshold), np.nextafter(2.0 * float(threshold), np.inf)))
    T = int(np.size(X))
    out = _gather(ctx, dt, Zb, nn.P, 1, idx, T)
    ZZ = ctx.download(out.ptr, np.shape(X), dt)
    ZZ[np.asarray(dists) > threshold * 2.0] = np.nan
    return ZZ


class _UpscalePlan:
    """fine regular grid -> model grid: kernel sizes + which fine node every model cell picks."""

    def __init__(self, X, Y, ctm_models_coordinate, grid_size, threshold):
        ctm_latitude = ctm_models_coordinate['Latitude']
        ctm_longitude = ctm_models_coordinate['Longitude']
        dlon = np.abs(ctm_longitude[0, 0] - ctm_longitude[0, 1])
        dlat = np.abs(ctm_latitude[0, 0] - ctm_latitude[1, 0])
        self.ctm_longitude = ctm_longitude
        self.ctm_latitude = ctm_latitude
        self.needed = bool((dlon >= grid_size) or (dlat >= grid_size))        # interpolator.py:64
        self.ties_resolved = 0
        self.Ny, self.Nx = np.shape(X)
        if not self.needed:
            return
        kx = np.floor(dlon / grid_size)
        ky = np.floor(dlat / grid_size)
        self.kx = 1 if kx == 0 else int(kx)
        self.ky = 1 if ky == 0 else int(ky)
        self.out_shape = np.shape(ctm_latitude)
        self.T = int(np.size(ctm_latitude))
        nn = NNIndex(X, Y)                                      # every fine node is a candidate (:78-82)
        self.idx, _ = nn.query_device(ctm_longitude, ctm_latitude, 2.0 * float(threshold))     # :83-91
        self.ties_resolved = nn.ties_resolved               # model centres equidistant from several fine nodes
        self.ctx = nn.ctx

    def run(self, fine_buf, nfields, dt, variance):
        """fine_buf: DeviceBuffer with nfields*(Ny*Nx) elements -> DeviceBuffer nfields*T"""
        out = self.ctx.alloc(nfields * self.T * dt.itemsize)
        self.ctx.check(self.ctx.lib.oisat_boxfilter_pick(self.ctx.h, _hip.dtype_code(dt), fine_buf.ptr, self.Ny, self.Nx,
                                                         nfields, self.ky, self.kx, 1 if variance else 0, self.idx.ptr,
                                                         self.T, out.ptr))
        return out


_plan_cache = {}


def _fingerprint(*arrays):
    key = []
    for a in arrays:
        a = np.asarray(a)
        key.append((a.shape, a.dtype.str, float(a.flat[0]), float(a.flat[-1]), float(a.sum()),
                    float(a.flat[a.size // 3]), float(a.flat[(2 * a.size) // 3])))
    return tuple(key)


def _upscale_plan(X, Y, ctm_models_coordinate, grid_size, threshold):
    key = _fingerprint(X, Y, ctm_models_coordinate['Latitude'], ctm_models_coordinate['Longitude']) + (
        float(grid_size), float(threshold))
    plan = _plan_cache.get(key)
    if plan is None:
        if len(_plan_cache) >= 8:
            _plan_cache.pop(next(iter(_plan_cache)))
        plan = _UpscalePlan(X, Y, ctm_models_coordinate, grid_size, threshold)
        _plan_cache[key] = plan
    return plan


def _upscaler(X: np.ndarray, Y: np.ndarray, Z: np.ndarray, ctm_models_coordinate: dict, grid_size: float,
              threshold: float, tri=None, error=False):
    """Drop-in for ``_upscaler`` (interpolator.py:48-97): when the model cells are at least as wide as the regridding cells,
    average the fine field over the footprint of a model cell (a box mean, or for ``error=True`` the variance of that mean)
    and pick the value at the fine node nearest to every model cell centre.

    ``X``, ``Y``, ``Z`` are the fine grid's coordinates and the field on it (2-D, same shape); ``ctm_models_coordinate`` holds
    the model's ``'Latitude'`` / ``'Longitude'`` meshes; ``grid_size`` is the fine spacing in degrees; model cells farther than
    ``threshold`` from every fine node come back NaN.  ``tri`` is accepted for signature compatibility and unused, as in the
    reference.  Returns ``(lon, lat, field, False)``, or the inputs with ``True`` when no upscaling is needed."""
    plan = _upscale_plan(X, Y, ctm_models_coordinate, grid_size, threshold)
    if not plan.needed:
        return X, Y, Z, True
    ctx = plan.ctx
    dt = _regrid_dtype()                  # convolve2d with a float64 kernel returns float64
    Zb = ctx.upload(Z, dtype=dt)
    out = plan.run(Zb, 1, dt, bool(error))
    Zc = ctx.download(out.ptr, plan.out_shape, dt)
    return plan.ctm_longitude, plan.ctm_latitude, Zc, False


_NOT_GIVEN = object()


class _QhullWorkers:
    """A few child processes that triangulate granules (``oisatgmi._qhull_worker``): qhull releases the GIL but the 0.5 s of
    ``Delaunay.transform`` (one LAPACK factorization per simplex) does not -- eight threads were slower than one -- so the
    host part of type 1 needs processes; plain children with pipes, one feeding thread each (the pickles are ~25 MB)."""

    def __init__(self, n):
        import subprocess
        import sys
        import threading
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        env["PYTHONPATH"] = os.pathsep.join([os.path.dirname(os.path.dirname(os.path.abspath(__file__)))] + [p for p in [env.get("PYTHONPATH")] if p])
        self.procs = [subprocess.Popen([sys.executable, "-m", "oisatgmi._qhull_worker"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                      for _ in range(int(n))]
        self.locks = [threading.Lock() for _ in self.procs]
        self.next = 0

    def submit(self, executor, lon, lat):
        """-> future of the triangulation (round-robin over the workers; ``executor``: a thread pool of one thread per worker)."""
        from . import _qhull_worker as w
        k = self.next % len(self.procs)
        self.next += 1

        def call():
            with self.locks[k]:
                p = self.procs[k]
                try:
                    w.write_msg(p.stdin, (np.asarray(lon), np.asarray(lat)))
                    return w.reply_of(p.stdout)
                except (w.WorkerDied, BrokenPipeError, OSError) as e:     # not "qhull failed": that is a reply
                    raise RuntimeError(f"triangulation worker {k} (pid {p.pid}) died, exit code {p.poll()}: {e}") from e
        return executor.submit(call)

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=10)
            except Exception:
                p.kill()


# --------------------------------------------------------------------------------------------
class _GranuleRegridder:
    """Everything ``interpolator()`` needs for one granule, resident in HBM."""

    def __init__(self, sat_data, grid_size, ctm_models_coordinate, flag_thresh, interpolator_type=4, triangulation=_NOT_GIVEN):
        """``triangulation``: the granule's ``scipy.spatial.Delaunay`` built ahead of time (``interpolator_many``), or None
        if qhull failed on it; by default it is built here."""
        self.ctx = ctx = _hip.context()
        self.kind = int(interpolator_type)
        self.rbf_memo = _RbfTies()      # type 3: what does not depend on the values is found with the first field stack only
        self.tri = None
        self.ok = True
        if self.kind == 1:
            if triangulation is _NOT_GIVEN:
                self.tri = TriIndex.from_points(sat_data.longitude_center, sat_data.latitude_center)
            else:
                self.tri = TriIndex(triangulation) if triangulation is not None else None
            if self.tri is None:            # qhull failed: the reference skips the granule (interpolator.py:151-155)
                self.ok = False
                return
        ctm_latitude = ctm_models_coordinate['Latitude']
        ctm_longitude = ctm_models_coordinate['Longitude']
        dlon = np.abs(ctm_longitude[0, 0] - ctm_longitude[0, 1])
        dlat = np.abs(ctm_latitude[0, 0] - ctm_latitude[1, 0])
        threshold_ctm = np.sqrt(dlon ** 2 + dlat ** 2)                                   # interpolator.py:121
        lat_min, lat_max = np.min(ctm_latitude), np.max(ctm_latitude)
        lon_min, lon_max = np.min(ctm_longitude), np.max(ctm_longitude)
        lon_grid = np.arange(lon_min, lon_max + grid_size, grid_size)                    # :141-143
        lat_grid = np.arange(lat_min, lat_max + grid_size, grid_size)
        self.lons_grid, self.lats_grid = np.meshgrid(lon_grid, lat_grid)
        self.fine_shape = self.lons_grid.shape
        self.Tfine = self.lons_grid.size
        self.P = int(np.size(sat_data.latitude_center))
        self.swath_shape = np.shape(np.squeeze(sat_data.quality_flag))
        self.flag_thresh = float(flag_thresh)
        self.qflag_host = np.squeeze(sat_data.quality_flag)
        self.nn = nn = NNIndex(sat_data.longitude_center, sat_data.latitude_center)
        self.cell = 2.0 * float(grid_size)
        # types 2 / 4 gather through this index (ties as the reference's tree breaks them); types 1 / 3 only use it as the
        # walk start / the `dists` mask, where a tie changes nothing
        self.idx_fine, _ = nn.query_device(self.lons_grid, self.lats_grid, self.cell,   # :145-150,:16-33
                                           resolve_ties=self.kind in (2, 4))
        self.plan = _upscale_plan(self.lons_grid, self.lats_grid, ctm_models_coordinate, grid_size, threshold_ctm)
        self._flag_bufs = {}
        if self.kind in (1, 3):
            self.tgt = ctx.alloc(2 * self.Tfine * 8)
            ctx.upload_into(self.tgt.at(0), np.ravel(self.lons_grid), dtype=np.float64)
            ctx.upload_into(self.tgt.at(self.Tfine * 8), np.ravel(self.lats_grid), dtype=np.float64)
        self.forced = None
        if self.kind == 1:              # once per granule; the reference repeats the search for every field
            self.forced = self.tri.locate(self.tgt, self.Tfine, self.idx_fine, self.lons_grid, self.lats_grid)

    def _flag(self, dt):
        b = self._flag_bufs.get(dt)
        if b is None:
            b = self._flag_bufs[dt] = self.ctx.upload(np.ravel(self.qflag_host), dtype=dt)
        return b

    def regrid(self, fields, error=False):
        """``fields``: list of swath-shaped arrays.  Returns (X, Y, Z, upscaled_ctm_needed) with Z of shape (fields, ny, nx),
        Z[f] = _upscaler(.., _interpolosis(tri, field*mask, ..), .., error=error)[2]."""
        ctx = self.ctx
        nf = len(fields)
        dt = _regrid_dtype()
        if error:                       # `uncertainty**2*mask`: the square happens in the field's own dtype
            fields = [np.asarray(a) ** 2 for a in fields]
        code = _hip.dtype_code(dt)
        item = dt.itemsize
        raw = ctx.alloc(nf * self.P * item)
        for f, a in enumerate(fields):
            a = np.squeeze(np.asarray(a))
            if a.size != self.P:
                raise ValueError(f"field {f} has {a.size} elements, swath has {self.P}")
            ctx.upload_into(raw.at(f * self.P * item), a, dtype=dt)
        masked = ctx.alloc(nf * self.P * item)
        flag = self._flag(dt)
        for f in range(nf):             # field*mask (x*1.0 | x*NaN), interpolator.py:126-128,:163
            ctx.check(ctx.lib.oisat_flag_mask(ctx.h, code, raw.at(f * self.P * item), flag.ptr, self.P, self.flag_thresh,
                                              0, masked.at(f * self.P * item)))
        if self.kind == 1:               # targets beyond 2*grid_size of any pixel carry idx -1 -> NaN, like the dists mask
            fine = self.tri.interpolate(dt, masked, nf, self.tgt, self.Tfine, self.idx_fine, self.forced)
        elif self.kind == 3:
            fine = _rbf(self.nn, dt, masked, nf, self.tgt, self.Tfine, self.idx_fine, self.cell, memo=self.rbf_memo)
        else:
            fine = _gather(ctx, dt, masked, self.P, nf, self.idx_fine, self.Tfine)
        if self.plan.needed:
            out = self.plan.run(fine, nf, dt, error)
            Z = ctx.download(out.ptr, (nf,) + tuple(self.plan.out_shape), dt)
            X, Y = self.plan.ctm_longitude, self.plan.ctm_latitude
            need = False
        else:
            Z = ctx.download(fine.ptr, (nf,) + tuple(self.fine_shape), dt)
            X, Y = self.lons_grid, self.lats_grid
            need = True
        return X, Y, Z, need            # Z[f] = field f (one array: consecutive fields are a cube without a copy)


def interpolator_many(interpolator_type: int, grid_size: float, granules, ctm_models_coordinate: dict, flag_thresh=0.75, workers=None):
    """``[interpolator(type, grid_size, g, ctm, flag_thresh) for g in granules]`` -- the loop the reference's readers run over
    a month's files (reader.py:1405, one joblib task per file; interpolator.py:151-159 builds the triangulation inside each)
    -- with the HOST part of type 1 taken off the critical path: the Delaunay triangulations of the granules ahead and their
    barycentric transforms (qhull 0.4-0.5 s + ``Delaunay.transform`` 0.5-0.6 s of a 0.85 s call for a 98,640-pixel OMI
    granule, against < 0.1 s of device work and copies) are built by ``workers`` host PROCESSES (default: the CPUs this
    process may use, at most 8) while the device regrids the current granule; the triangulation comes back pickled (23 MB,
    30 ms).  Same triangulations, same order of evaluation: the outputs are those of the serial calls bit for bit (``None``
    entries and ``None`` results as there).  The workers are child processes (``oisatgmi._qhull_worker``, pipes) that never
    touch the device; only this process holds the handle (INTEGRATION.md "Threading / processes")."""
    granules = list(granules)
    if interpolator_type != 1 or len(granules) < 2:
        return [None if g is None else interpolator(interpolator_type, grid_size, g, ctm_models_coordinate, flag_thresh) for g in granules]
    from concurrent.futures import ThreadPoolExecutor
    if workers is None:
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        workers = max(1, min(8, ncpu - 1, len(granules)))
    ahead = 2 * int(workers)                              # triangulations in flight / waiting: bounded (each holds ~25 MB)
    out = []
    procs = _QhullWorkers(workers)
    try:
        with ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix="oisat-qhull") as feeders:
            futures = {}

            def submit(k):
                g = granules[k]
                if g is not None:
                    futures[k] = procs.submit(feeders, g.longitude_center, g.latitude_center)

            for k in range(min(ahead, len(granules))):
                submit(k)
            for k, g in enumerate(granules):
                if k + ahead < len(granules):
                    submit(k + ahead)
                if g is None:
                    out.append(None)
                    continue
                tri = futures.pop(k).result()
                out.append(_interpolate_granule(interpolator_type, grid_size, g, ctm_models_coordinate, flag_thresh, tri))
    finally:
        procs.close()
    return out


def interpolator(interpolator_type: int, grid_size: float, sat_data, ctm_models_coordinate: dict, flag_thresh=0.75):
    """Drop-in for ``interpolator`` (interpolator.py:100-291): one satellite granule onto the model grid.

    ``interpolator_type`` picks the scheme between swath pixels and the regular ``grid_size``-degree grid laid over the model
    region: 1 = linear on a Delaunay triangulation (what the reference recommends), 2 = nearest pixel, 3 = thin-plate-spline
    RBF on the five nearest pixels, 4 = nearest pixel by k-d tree (same answer as 2).  ``sat_data`` is a ``satellite_amf`` or
    ``satellite_opt`` record; pixels whose quality flag is at or below ``flag_thresh`` are masked before regridding.
    ``ctm_models_coordinate`` holds the model's ``'Latitude'`` / ``'Longitude'`` meshes.  Returns a record of the same type on
    the model grid, or ``None`` when the granule misses the region or cannot be triangulated."""
    return _interpolate_granule(interpolator_type, grid_size, sat_data, ctm_models_coordinate, flag_thresh, _NOT_GIVEN)


def _interpolate_granule(interpolator_type, grid_size, sat_data, ctm_models_coordinate, flag_thresh, triangulation):
    if interpolator_type not in (1, 2, 3, 4):
        raise Exception("other type of interpolation methods has not been implemented yet")
    rg = _GranuleRegridder(sat_data, grid_size, ctm_models_coordinate, flag_thresh, interpolator_type, triangulation)
    if not rg.ok:
        return None
    is_amf = isinstance(sat_data, satellite_amf)
    is_opt = isinstance(sat_data, satellite_opt)

    # ---- one stacked pass for every mean-kernel field of the granule
    names, fields = ["vcd"], [sat_data.vcd]
    print('....................... vcd')
    if is_amf:
        print('....................... amf')
        names.append("amf")
        fields.append(sat_data.amf)
    print('....................... tropopause')
    has_trop = np.size(sat_data.tropopause) != 1
    if has_trop:
        names.append("tropopause")
        fields.append(sat_data.tropopause)
    nz = np.shape(sat_data.pressure_mid)[0]
    levels = {}                                           # name -> (first index, count)

    def add_levels(name, cube, count):
        levels[name] = (len(fields), count)
        for z in range(count):
            names.append(f"{name}[{z}]")
            fields.append(np.squeeze(np.asarray(cube)[z]))

    if is_amf and np.size(sat_data.scattering_weights) != 1:
        print('....................... SWs [' + str(nz) + ' levels]')
        add_levels("scattering_weights", sat_data.scattering_weights, nz)
        print('....................... pmids [' + str(nz) + ' levels]')
        add_levels("pressure_mid", sat_data.pressure_mid, nz)
    if is_opt:
        singles = {}
        for nm, msg in (("aprior_column", "apriori column"), ("surface_pressure", "surface pressure"),
                        ("apriori_surface", "apriori surface")):
            if getattr(sat_data, nm).any():
                print('....................... ' + msg)
                singles[nm] = len(fields)
                names.append(nm)
                fields.append(getattr(sat_data, nm))
        print('....................... Xcol')
        singles["x_col"] = len(fields)
        names.append("x_col")
        fields.append(sat_data.x_col)
        if sat_data.sensor == 'MOPITT':
            add_levels("averaging_kernels", sat_data.averaging_kernels, nz + 1)
        if sat_data.sensor == 'GOSAT':
            add_levels("averaging_kernels", sat_data.averaging_kernels, nz)
            add_levels("pressure_weight", sat_data.pressure_weight, nz)
        add_levels("pressure_mid", sat_data.pressure_mid, nz)
        add_levels("apriori_profile", sat_data.apriori_profile, nz)

    upscaled_X, upscaled_Y, Z, upscaled_ctm_needed = rg.regrid(fields)
    vcd = Z[0]
    with np.errstate(all="ignore"):
        if np.isnan(vcd).all():
            print("the satellite granule doesn't fall into the region - skipping!")
            return None
    by_name = dict(zip(names, Z))

    def cube(name):                     # the levels were regridded as consecutive fields: their block of Z is the cube
        first, count = levels[name]     # (np.stack of 35 global float64 fields was 21 ms a cube, half of a type-4 call)
        return Z[first:first + count].astype(np.float64, copy=False)

    tropopause = by_name["tropopause"] if has_trop else np.empty((1))
    latitude_center = upscaled_Y
    longitude_center = upscaled_X
    print('....................... error')
    uncertainty = rg.regrid([sat_data.uncertainty], error=True)[2][0]             # variance kernel, :185-187
    uncertainty = np.sqrt(uncertainty)                                           # :188

    if is_amf:
        if "scattering_weights" in levels:
            scattering_weights = cube("scattering_weights")
            pressure_mid = cube("pressure_mid")
        else:
            scattering_weights = np.empty((1))
            pressure_mid = np.zeros((nz, np.shape(upscaled_X)[0], np.shape(upscaled_X)[1]))
        return satellite_amf(vcd, by_name["amf"], sat_data.time, tropopause, latitude_center, longitude_center, [], [],
                             uncertainty, [], pressure_mid, scattering_weights, upscaled_ctm_needed, [], [], [], [])
    if is_opt:
        missing = [nm for nm in ("aprior_column", "surface_pressure", "apriori_surface") if nm not in singles]
        if missing:                      # the reference leaves these names unbound and dies at :285-287
            raise NameError(f"satellite_opt record has all-zero {missing}; the reference cannot rebuild it either")
        if sat_data.sensor == 'MOPITT':
            pressure_weights = np.empty((1))
        elif sat_data.sensor == 'GOSAT':
            pressure_weights = cube("pressure_weight")
        else:
            raise NameError("averaging_kernels are only regridded for sensor 'MOPITT' or 'GOSAT'")
        return satellite_opt(vcd, sat_data.time, [], tropopause, latitude_center, longitude_center, [], [],
                             uncertainty, [], cube("pressure_mid"), cube("averaging_kernels"), upscaled_ctm_needed,
                             [], [], [], by_name["aprior_column"], cube("apriori_profile"), by_name["surface_pressure"],
                             by_name["apriori_surface"], by_name["x_col"], pressure_weights, sat_data.sensor)
    raise TypeError("sat_data must be a satellite_amf or satellite_opt record")
