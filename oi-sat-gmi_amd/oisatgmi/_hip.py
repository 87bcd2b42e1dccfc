"""ctypes binding of ``liboisat_hip.so`` (C-ABI: ``include/oisat.h``).

There is deliberately NO CPU fallback in this package: if the shared library is missing, was
not built for this machine, or no gfx950 device is visible, every compute entry point raises
``OisatUnavailable`` with the reason.  (The float64 NumPy oracle under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import collections
import ctypes as C
import os
import sys
import threading

import numpy as np

F32, F64 = 0, 1
MAX_SCALES = 128

_c_ctx = C.c_void_p
_i64 = C.c_int64
_ptr = C.c_void_p

# name -> (restype, argtypes); mirrors include/oisat.h one to one (tests check the symbol list)
SIGNATURES = {
    "oisat_init": (C.c_int, [C.c_int, C.POINTER(_c_ctx)]),
    "oisat_shutdown": (None, [_c_ctx]),
    "oisat_last_error": (C.c_char_p, []),
    "oisat_version": (C.c_char_p, []),
    "oisat_device_info": (C.c_int, [_c_ctx, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(_i64)]),
    "oisat_set_stream": (C.c_int, [_c_ctx, _ptr]),
    "oisat_stream_create": (C.c_int, [_c_ctx]),
    "oisat_bind_thread": (C.c_int, [_c_ctx]),
    "oisat_wait_for": (C.c_int, [_c_ctx, _c_ctx]),
    "oisat_query": (C.c_int, [_c_ctx, C.POINTER(C.c_int)]),
    "oisat_stream_create_masked": (C.c_int, [_c_ctx, C.c_int]),
    "oisat_set_share": (C.c_int, [_c_ctx, C.c_int, C.c_int]),
    "oisat_set_refine_tol": (C.c_int, [_c_ctx, C.c_double]),
    "oisat_sync": (C.c_int, [_c_ctx]),
    "oisat_dmalloc": (C.c_int, [_c_ctx, C.c_size_t, C.POINTER(_ptr)]),
    "oisat_dfree": (C.c_int, [_c_ctx, _ptr]),
    "oisat_h2d": (C.c_int, [_c_ctx, _ptr, _ptr, C.c_size_t]),
    "oisat_d2h": (C.c_int, [_c_ctx, _ptr, _ptr, C.c_size_t]),
    "oisat_memset": (C.c_int, [_c_ctx, _ptr, C.c_int, C.c_size_t]),
    "oisat_prof_enable": (C.c_int, [_c_ctx, C.c_int]),
    "oisat_prof_reset": (C.c_int, [_c_ctx]),
    "oisat_prof_collect": (C.c_int, [_c_ctx, C.c_int, _ptr, C.POINTER(C.c_double), C.POINTER(_i64)]),
    "oisat_oi_curve": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, C.POINTER(C.c_double), C.c_int,
                                 C.POINTER(C.c_double), C.POINTER(_i64)]),
    "oisat_oi_apply": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _ptr, _ptr, _i64, C.c_double, _ptr, _ptr, _ptr, _ptr]),
    "oisat_oi_fused": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _ptr, _ptr, _i64, C.POINTER(C.c_double), C.c_int, C.c_int,
                                 _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "oisat_nanmean_stack": (C.c_int, [_c_ctx, C.c_int, _ptr, C.c_int, _i64, C.c_int, _ptr]),
    "oisat_error_average": (C.c_int, [_c_ctx, C.c_int, _ptr, C.c_int, _i64, C.c_int, _ptr]),
    "oisat_affine": (C.c_int, [_c_ctx, C.c_int, _ptr, _i64, C.c_double, C.c_double, _ptr]),
    "oisat_oi_variances": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, C.c_double, _ptr, _ptr]),
    "oisat_scaling_factor": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr]),
    "oisat_partial_column": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr]),
    "oisat_amf_recal": (C.c_int, [_c_ctx, _ptr, _ptr, C.c_int, C.c_int, _ptr, _ptr, C.c_int, _ptr, _ptr, _ptr, _i64, _ptr, _ptr,
                                  _ptr]),
    "oisat_column_sum": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, C.c_int, _ptr, _ptr, _i64, _ptr]),
    "oisat_ak_conv_mopitt": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _ptr, C.c_int, _ptr, _ptr, _ptr, C.c_int, _ptr, _ptr, _ptr, _i64,
                                       _ptr, _ptr]),
    "oisat_ak_conv_gosat": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, C.c_int, _ptr, _ptr, _ptr, _ptr, C.c_int, _ptr, _i64, _ptr]),
    "oisat_water_column": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr]),
    "oisat_pwv_sum": (C.c_int, [_c_ctx, C.c_int, _ptr, C.c_int, _ptr, _i64, _ptr]),
    "oisat_boxfilter_symm": (C.c_int, [_c_ctx, C.c_int, _ptr, _i64, _i64, C.c_int, C.c_int, C.c_int, _ptr]),
    "oisat_nn_query": (C.c_int, [_c_ctx, _ptr, _ptr, _i64, _ptr, _ptr, _i64, C.c_double, _ptr, _ptr]),
    "oisat_nn_query_ties": (C.c_int, [_c_ctx, _ptr, _ptr, _i64, _ptr, _ptr, _i64, C.c_double, _ptr, _ptr, _ptr,
                                      C.POINTER(C.c_int64)]),
    "oisat_gather_mask": (C.c_int, [_c_ctx, C.c_int, _ptr, _i64, C.c_int, _ptr, _i64, _ptr]),
    "oisat_linear_interp": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64,
                                      C.c_int, _ptr, C.POINTER(C.c_double)]),
    "oisat_linear_locate": (C.c_int, [_c_ctx, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _i64, C.POINTER(C.c_double),
                                      _ptr, C.POINTER(C.c_int64)]),
    "oisat_linear_interp_forced": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _ptr, _ptr, _i64, _ptr, _i64,
                                             C.c_int, _ptr, C.POINTER(C.c_double), _ptr]),
    "oisat_rbf_interp": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, C.c_double, C.c_int, _ptr, C.c_int,
                                   _ptr, C.POINTER(C.c_int64)]),
    "oisat_tri_transform": (C.c_int, [_c_ctx, _ptr, _i64, _ptr, _i64, _ptr, _ptr, C.POINTER(C.c_int64)]),
    "oisat_rbf_interp_ties": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, C.c_double, C.c_int, _ptr, C.c_int,
                                        _ptr, C.POINTER(C.c_int64), _ptr, C.POINTER(C.c_int64)]),
    "oisat_rbf_interp_forced": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, _ptr, _i64, C.c_int, _ptr,
                                          C.c_int, _ptr, C.POINTER(C.c_int64)]),
    "oisat_rbf_check_masked": (C.c_int, [_c_ctx, _ptr, _ptr, _i64, _ptr, _ptr, _i64, _ptr, C.c_double, C.c_int, C.POINTER(_i64)]),
    "oisat_boxfilter_pick": (C.c_int, [_c_ctx, C.c_int, _ptr, _i64, _i64, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _ptr, _i64, _ptr]),
    "oisat_flag_mask": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, C.c_double, C.c_int, _ptr]),
    "oisat_sqrt": (C.c_int, [_c_ctx, C.c_int, _ptr, _i64, _ptr]),
    "oisat_cov_build": (C.c_int, [_c_ctx, _ptr, _ptr, _ptr, _i64, C.c_double, _ptr, _i64]),
    "oisat_innovation": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _ptr, _i64, _ptr]),
    "oisat_gemm_nt": (C.c_int, [_c_ctx, _ptr, _i64, _ptr, _i64, _ptr, _i64, _i64, _i64, _i64, C.c_int, C.c_int]),
    "oisat_potrf": (C.c_int, [_c_ctx, _ptr, _i64, _i64, C.POINTER(C.c_int)]),
    "oisat_potrs": (C.c_int, [_c_ctx, _ptr, _i64, _i64, _ptr]),
    "oisat_cov_residual": (C.c_int, [_c_ctx, _ptr, _ptr, _ptr, _i64, C.c_double, _ptr, _ptr, _ptr, _ptr]),
    "oisat_gain_solve": (C.c_int, [_c_ctx, _ptr, _ptr, _ptr, _ptr, _i64, _i64, C.c_double, _ptr, C.c_int, _ptr,
                                   C.POINTER(C.c_double), _ptr]),
    "oisat_trsm_rows": (C.c_int, [_c_ctx, _ptr, _i64, _i64, _ptr, _i64, _i64]),
    "oisat_posterior_error": (C.c_int, [_c_ctx, _ptr, _i64, _i64, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, C.c_double, _i64, _ptr]),
    "oisat_gain_diag": (C.c_int, [_c_ctx, _ptr, _i64, _i64, _ptr, _i64, _ptr]),
    "oisat_apply_increment": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _ptr, _ptr, _ptr, _i64, C.c_double, _ptr,
                                        _ptr, _ptr, _ptr, _ptr]),
    "oisat_apply_increment_grid": (C.c_int, [_c_ctx, C.c_int, _ptr, _ptr, _i64, _i64, _ptr, _ptr, _ptr, _i64, C.c_double, _ptr,
                                             _ptr, _ptr, _ptr, _ptr]),
    "oisat_batch_create": (C.c_int, [_c_ctx, C.c_int, C.POINTER(_ptr), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_ptr),
                                     C.POINTER(C.c_int)]),
    "oisat_batch_potrf": (C.c_int, [_c_ctx, C.c_int, C.POINTER(C.c_int)]),
    "oisat_batch_set_solve": (C.c_int, [_c_ctx, C.c_int, C.c_int] + [C.POINTER(_ptr)] * 11 + [C.POINTER(_i64)] + [C.POINTER(_ptr)] * 3),
    "oisat_set_task_graph": (C.c_int, [_c_ctx, C.c_int]),
    "oisat_dag_task_order": (C.c_int, [C.c_int, _ptr, C.c_int, _ptr, _i64, C.POINTER(_i64), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "oisat_batch_set_grid": (C.c_int, [_c_ctx, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_ptr)]),
    "oisat_set_obs_blocks": (C.c_int, [_c_ctx, _ptr, _i64]),
    "oisat_batch_solve": (C.c_int, [_c_ctx, C.c_int, C.c_int, C.c_double, C.c_int]),
    "oisat_batch_analyse": (C.c_int, [_c_ctx, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]),
    "oisat_batch_is_task_graph": (C.c_int, [_c_ctx, C.c_int, C.POINTER(C.c_int)]),
    "oisat_batch_destroy": (C.c_int, [_c_ctx, C.c_int]),
    "oisat_factor_adopt": (C.c_int, [_c_ctx, _ptr, _i64, _i64, _ptr]),
    "oisat_comm_unique_id": (C.c_int, [C.c_char_p, C.c_int]),
    "oisat_comm_init": (C.c_int, [_c_ctx, C.c_int, C.c_int, C.c_char_p]),
    "oisat_comm_bcast": (C.c_int, [_c_ctx, _ptr, C.c_size_t, C.c_int]),
    "oisat_comm_gather": (C.c_int, [_c_ctx, _ptr, C.c_size_t, _ptr, C.c_int]),
    "oisat_comm_destroy": (C.c_int, [_c_ctx]),
    "oisat_solve_status": (C.c_int, [_c_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "oisat_solve_status_ex": (C.c_int, [_c_ctx, C.POINTER(C.c_int32), C.c_int, C.c_int]),
    "oisat_dense_reserve": (C.c_int, [_c_ctx, _i64, _i64]),
}


class OisatError(RuntimeError):
    """A call into liboisat_hip.so returned an error code."""


class OisatUnavailable(RuntimeError):
    """The HIP backend cannot run here (library missing / no MI355X).  There is no fallback."""


def library_path() -> str:
    env = os.environ.get("OISAT_LIB")
    if env:
        return env
    here = os.path.dirname(os.path.abspath(__file__))
    return os.path.join(os.path.dirname(here), "lib", "liboisat_hip.so")


_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen the library and declare every prototype.  Works without a GPU (no compute call)."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        path = library_path()
        if not os.path.exists(path):
            raise OisatUnavailable(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C oi-sat-gmi_amd/csrc`.  This package has no CPU fallback.")
        # Independent tiles / months run on several handles, one stream each (dense.TiledAnalysis).  The HIP
        # runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default: with 12 lanes the
        # tiled 720x1440 analysis takes 0.53 s on 4 queues and 0.47 s on 16.  Read once, when the runtime
        # initialises, so this only has an effect if nothing in the process has touched the GPU yet.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
        # PyTorch wheels bundle their own HIP runtime.  If this library pulls in /opt/rocm's copy first
        # and torch is imported afterwards, torch finds "No HIP GPUs"; the other order is fine.  So when
        # torch is installed, let it load its runtime first (OISAT_PRELOAD_TORCH=0 disables this).
        if "torch" not in sys.modules and os.environ.get("OISAT_PRELOAD_TORCH", "1") != "0":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        try:
            lib = C.CDLL(path)
        except OSError as e:
            raise OisatUnavailable(f"cannot load {path}: {e}.  This package has no CPU fallback.") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header/library drift
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def load_test_hooks_library():
    """The -DOISAT_TEST_HOOKS build of the library (csrc/Makefile: fault injection into the task-graph launch, read from
    OISAT_DAG_FLAGS there and nowhere else).  For tests/test_gpu_dag.py: ``Context(device, lib=load_test_hooks_library())``;
    the product never loads it."""
    load_library()                                          # (torch's HIP runtime first, as above)
    path = os.path.join(os.path.dirname(library_path()), "liboisat_hip_testhooks.so")
    if not os.path.exists(path):
        raise OisatUnavailable(f"{path} not found: `make -C oi-sat-gmi_amd/csrc` builds it next to the product library")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def dtype_code(dt) -> int:
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError(f"unsupported field dtype {dt}; float32 or float64 only")


def compute_dtype(*arrays) -> np.dtype:
    """Field dtype policy of the drop-in surface: follow the inputs like NumPy would (any float64
    or non-float input -> float64, all float32 -> float32).  ``OISAT_DTYPE=f32|f64`` overrides."""
    env = os.environ.get("OISAT_DTYPE", "").lower()
    if env in ("f32", "float32"):
        return np.dtype(np.float32)
    if env in ("f64", "float64"):
        return np.dtype(np.float64)
    for a in arrays:
        if np.asarray(a).dtype != np.float32:
            return np.dtype(np.float64)
    return np.dtype(np.float32)


class SolveStatus(collections.namedtuple("SolveStatus", "notpd_col notpd_blocks trsv_timeouts unconverged unconverged_member dag_timeouts")):
    """include/oisat.h ``OISAT_STATUS_*``; ``clean``: nothing to report."""
    __slots__ = ()

    @property
    def clean(self) -> bool:
        return not (self.notpd_col or self.notpd_blocks or self.trsv_timeouts or self.unconverged or self.dag_timeouts)


class DeviceBuffer:
    """A block of HBM owned through the C-ABI (``oisat_dmalloc``/``oisat_dfree``)."""

    __slots__ = ("ctx", "ptr", "nbytes", "foreign")

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.foreign = False
        cached = ctx._take_cached(self.nbytes)
        if cached is not None:
            self.ptr = cached
            return
        p = _ptr()
        ctx.check(ctx.lib.oisat_dmalloc(ctx.h, max(self.nbytes, 16), C.byref(p)))
        self.ptr = p.value

    def shared_with_other_streams(self):
        """Mark a buffer that streams other than its handle's also touch (a factor the BatchedFactor group streams work on,
        a slab torch's RCCL stream reads): it is then never parked in the handle's reuse cache -- which orders a reused
        buffer only behind the OWNING stream -- but released through ``hipFree``, which waits for the whole device."""
        self.foreign = True
        return self

    def free(self):
        if self.ptr is not None and self.ctx is not None and self.ctx.h is not None:
            if self.foreign or not self.ctx._give_cached(self.nbytes, self.ptr):
                self.ctx.lib.oisat_dfree(self.ctx.h, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def at(self, byte_offset: int) -> int:
        return self.ptr + int(byte_offset)


class Context:
    """One handle per process and device: ``oisat_init`` + the handle's stream."""

    def __init__(self, device: int, lib=None):
        self.lib = lib if lib is not None else load_library()
        h = _c_ctx()
        rc = self.lib.oisat_init(int(device), C.byref(h))
        if rc != 0:
            msg = self.lib.oisat_last_error().decode()
            raise OisatUnavailable(f"oisat_init(device={device}) failed: {msg}.  This package has no CPU fallback.")
        self.h = h
        self.device = int(device)
        # Freed buffers of up to 256 MB are parked (2 GB at most) and handed out again for a request of exactly the same
        # size: the drop-in calls (OI, averaging, interpolator) allocate the same handful of sizes on every call, and a
        # hipMalloc / hipFree pair costs more than the kernels they bracket.  A parked buffer is ordered behind its OWN
        # handle's stream only, so buffers that other streams touch opt out (DeviceBuffer.shared_with_other_streams).
        self._cache = {}
        self._cache_bytes = 0
        self._cache_lock = threading.Lock()

    _CACHE_MAX_BUFFER = 256 << 20
    _CACHE_MAX_TOTAL = 2 << 30

    def _take_cached(self, nbytes):
        with self._cache_lock:
            lst = self._cache.get(nbytes)
            if lst:
                self._cache_bytes -= nbytes
                return lst.pop()
        return None

    def _give_cached(self, nbytes, ptr) -> bool:
        if nbytes <= 0 or nbytes > self._CACHE_MAX_BUFFER or os.environ.get("OISAT_BUFFER_CACHE", "1") == "0":
            return False
        try:
            self.sync()                  # what hipFree would have waited for on this handle's stream
        except OisatError:
            return False
        with self._cache_lock:
            if self._cache_bytes + nbytes > self._CACHE_MAX_TOTAL:
                return False
            self._cache.setdefault(nbytes, []).append(ptr)
            self._cache_bytes += nbytes
        return True

    def drop_cache(self):
        with self._cache_lock:
            for lst in self._cache.values():
                for ptr in lst:
                    self.lib.oisat_dfree(self.h, ptr)
            self._cache.clear()
            self._cache_bytes = 0

    # ---- plumbing
    def check(self, rc: int):
        if rc != 0:
            raise OisatError(f"liboisat_hip error {rc}: {self.lib.oisat_last_error().decode()}")

    def close(self):
        if self.h is not None:
            self.drop_cache()
            self.lib.oisat_shutdown(self.h)
            self.h = None

    def set_stream(self, stream_handle: int | None):
        self.check(self.lib.oisat_set_stream(self.h, stream_handle or None))

    def own_stream(self):
        """Give this handle a stream of its own (used for concurrent tiles)."""
        self.check(self.lib.oisat_stream_create(self.h))
        return self

    def own_stream_masked(self, reserve_per_xcd: int):
        """A stream of its own whose kernels keep off ``reserve_per_xcd`` CUs of every XCD."""
        self.check(self.lib.oisat_stream_create_masked(self.h, int(reserve_per_xcd)))
        return self

    def bind_thread(self):
        """Make this handle's device the calling thread's current HIP device (per-thread state in HIP)."""
        self.check(self.lib.oisat_bind_thread(self.h))

    def wait_for(self, other: "Context"):
        """Order this handle's stream after everything enqueued so far on ``other``'s stream (device-side)."""
        self.check(self.lib.oisat_wait_for(self.h, other.h))

    def sync(self):
        self.check(self.lib.oisat_sync(self.h))

    def busy(self) -> bool:
        """True while work enqueued on this handle's stream is still in flight (``hipStreamQuery``)."""
        b = C.c_int(0)
        self.check(self.lib.oisat_query(self.h, C.byref(b)))
        return bool(b.value)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def upload(self, arr: np.ndarray, dtype=None) -> DeviceBuffer:
        a = np.ascontiguousarray(arr, dtype=dtype)
        buf = DeviceBuffer(self, a.nbytes)
        self.check(self.lib.oisat_h2d(self.h, buf.ptr, a.ctypes.data, a.nbytes))
        self.sync()                      # `a` may be a temporary: finish the copy before it dies
        return buf

    def upload_into(self, dev_ptr: int, arr: np.ndarray, dtype=None) -> int:
        a = np.ascontiguousarray(arr, dtype=dtype)
        self.check(self.lib.oisat_h2d(self.h, dev_ptr, a.ctypes.data, a.nbytes))
        self.sync()
        return a.nbytes

    def download(self, dev_ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.oisat_d2h(self.h, out.ctypes.data, dev_ptr, out.nbytes))
        return out

    def device_info(self):
        name = C.create_string_buffer(128)
        cu = C.c_int()
        hbm = _i64()
        self.check(self.lib.oisat_device_info(self.h, name, 128, C.byref(cu), C.byref(hbm)))
        return {"name": name.value.decode(), "cu_count": cu.value, "hbm_bytes": hbm.value}

    def solve_status(self, clear=True) -> "SolveStatus":
        """The six status words of ``oisat_solve_status_ex`` since the last clear."""
        w = (C.c_int32 * len(SolveStatus._fields))()
        self.check(self.lib.oisat_solve_status_ex(self.h, w, len(SolveStatus._fields), 1 if clear else 0))
        return SolveStatus(*(int(x) for x in w))

    def check_solves(self, what="dense analysis"):
        """Raise ``OisatError`` if any unchecked dense solve on this handle failed since the last check: a non-positive
        pivot, a time-out of the task graph or of a triangular sweep, or a refinement that used all its rounds and ended
        above its tolerance."""
        st = self.solve_status(clear=True)._asdict()
        msgs = []
        if st["notpd_col"] or st["notpd_blocks"]:
            msgs.append(f"H B H^T + R not positive definite (first bad column {st['notpd_col']}, {st['notpd_blocks']} diagonal block(s))")
        if st["dag_timeouts"]:
            msgs.append(f"{st['dag_timeouts']} task-graph factorization(s) timed out: incomplete factor")
        if st["trsv_timeouts"]:
            msgs.append(f"{st['trsv_timeouts']} triangular-solve workgroup(s) gave up waiting: z holds NaN fill")
        if st["unconverged"]:
            who = f", first: batch member {st['unconverged_member']}" if st["unconverged_member"] >= 0 else ""
            msgs.append(f"{st['unconverged']} gain solve(s) ended above the refinement tolerance after every allowed round{who} "
                        "(raise `refine`, or the observation error / lower the correlation length: the fp32 factor is a poor preconditioner)")
        if msgs:
            raise OisatError(f"{what}: " + "; ".join(msgs))

    # ---- profiling
    def prof_enable(self, on=True):
        self.check(self.lib.oisat_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self.check(self.lib.oisat_prof_reset(self.h))

    def prof_collect(self):
        cap = 64
        names = (C.c_char * 64 * cap)()
        ms = (C.c_double * cap)()
        cnt = (_i64 * cap)()
        n = self.lib.oisat_prof_collect(self.h, cap, C.cast(names, _ptr), ms, cnt)
        if n < 0:
            self.check(n)
        out = {}
        for i in range(min(n, cap)):
            out[bytes(names[i]).split(b"\0", 1)[0].decode()] = {"total_ms": ms[i], "launches": cnt[i]}
        return out


_ctx = None
_ctx_lock = threading.Lock()


def default_device() -> int:
    for var in ("OISAT_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v)
    return 0


def context() -> Context:
    """The process-wide handle (created on first use).  Raises ``OisatUnavailable`` loudly when the
    HIP backend cannot run -- by design nothing in this package computes on the CPU instead."""
    global _ctx
    with _ctx_lock:
        if _ctx is None:
            _ctx = Context(default_device())
        return _ctx


def reset_context():
    global _ctx
    with _ctx_lock:
        if _ctx is not None:
            _ctx.close()
            _ctx = None
